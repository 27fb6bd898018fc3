// nl_rows.hpp -- the fine-row pair search (fp32, open box): the default COUNT sweep and expansion of round 3.
//
// Why.  The 27-cell search of nl_kernels.hpp / nl_lean.hpp tests every particle of a cell against all 27 cells of its
// stencil: 27 <N/cell> candidates for a sphere that holds 4.4 cells' worth of them.  Here the cell-sorted array is
// also sorted by the QUARTER of the cell a particle lies in along z (k_bin_cells<FINE>): a row of x-cells becomes four
// fine rows, each contiguous in x.  A particle in quarter q of its cell can only reach the quarter-planes q - 4 .. q + 4
// of the twelve the stencil holds along z -- 9 of 12, 75 % of the candidates -- because two particles whose quarter
// indices differ by 5 are more than one cell edge >= rc apart along z.
//
// How.  One workgroup per cell, one wave per quarter of the cell.  The 36 fine-row windows of the stencil (12 quarter
// planes x 3 rows of x-cells, each window 3 x-cells: contiguous in the sorted array) are staged back to back in LDS,
// plane by plane, so the 27 windows of wave q are ONE contiguous piece of the stream, [off[3 q], off[3 q + 27]).  The
// wave walks it in 64-slot tiles: lane l of tile g tests stream slot s1 + 64 g + l and bit g of the lane's hit word
// stands for that slot.  No index table, no gather list, no branch in the tile loop; the expansion finds the id of
// bit g of lane l at ids[s1 + 64 g + l].  (The last tile may run past the wave's piece into the next plane: those are
// particles of the stencil out of the wave's reach, tested like any other and, by the argument below, never accepted.)
//
// Same arithmetic, same visit rule: a pair is tested only if its cells are stencil neighbours in the reference's sense
// (MakeNeighMeshId neighlist_gpu.hpp:125-142 / neighlist_cpu.hpp:107-132, periodic wrap of the cell index), with
// r2 = (dx*dx + dy*dy) + dz*dz rounded operation by operation, kept unless r2 > rc2 (neighlist_cpu.hpp:219-223), owner
// rule of RegistInteractPair (neighlist_cpu.hpp:225-236).  Pairs that are NOT tested here but are by the reference lie
// five or more quarter-planes apart: the rounded products t = z * ims of the two particles differ by more than 1, so
// their distance along z exceeds ms (1 - 8 m 2^-24); the host takes this path only where ms >= rc (1 + 8 m 2^-24 +
// 2^-20) along z (rows_margin_ok in nl_api.hip), i.e. where such a pair fails the cut-off test in any rounding.
// Boxes whose cell edge equals the cut-off to within that margin (Lz / rc an integer) take the 27-cell path.
//
// Kernels: k_sweep_rows_f32 (COUNT: counts + hit words), k_fill_rows (the list from the hit words), k_rows_overflow
// (cells whose stream does not fit the LDS buffer: searched straight from memory, both passes; rare).
#pragma once

namespace nl {

#ifndef NL_ROWS_SHARES   // 1: the cell's particles in four equal shares; 0: wave w of the COUNT sweep takes quarter w
#define NL_ROWS_SHARES 1
#endif
// timing experiments only (wrong lists; profiles/r03_fine_rows_investigation.txt)
#ifndef NL_ROWS_DIAG  // 1 = no hit-word stores, 2 = no count stores, 4 = no tile loop
#define NL_ROWS_DIAG 0
#endif
#ifndef NL_ROWS_EXIT  // the COUNT sweep leaves after 1 = its first instruction, 2 = the window table, 3 = the DMA issue, 4 = the barrier
#define NL_ROWS_EXIT 0
#endif
#ifndef NL_ROWS_GATHER   // 0: the stream is staged window by window
#define NL_ROWS_GATHER 1
#endif

constexpr int ROWS_WAVES = 4;  // one wave per quarter of the i-cell
constexpr int ROWS_WIN = 36;   // windows of a cell's stencil: slot = wz * 3 + (dy + 1), wz = (dz + 1) * 4 + quarter
constexpr int ROWS_SPAN = 27;  // windows a wave walks: quarter planes q .. q + 8
#ifndef NL_ROWS_G
#define NL_ROWS_G 12
#endif
constexpr int ROWS_G = NL_ROWS_G;  // most i-particles of a group (even, <= 12)

// V: 0 = 16-bit hit words, the LDS stream of 8 workgroups per CU; 1, 2: 32-bit words for denser boxes (6 and 4
// workgroups per CU).  CAP: staged particles per cell (+ 1 sentinel); a wave's piece may span word bits x 64 slots.
// WS / WF: resident waves per SIMD the LDS of the COUNT sweep / of the expansion allows.
template <int V> struct RowsCfg;
template <> struct RowsCfg<0> { static constexpr int CAP = 1279, WS = 8, WF = 8; typedef uint16_t word_t; };
template <> struct RowsCfg<1> { static constexpr int CAP = 1663, WS = 6, WF = 6; typedef uint32_t word_t; };
template <> struct RowsCfg<2> { static constexpr int CAP = 2495, WS = 4, WF = 5; typedef uint32_t word_t; };

struct RowsArgs {
  const Pos<float>* __restrict__ sorted;
  const int32_t* __restrict__ sorted_row;
  const int32_t* __restrict__ sorted_gid;
  const int32_t* __restrict__ fine_start;  // rows_fine_index: [(row * mx + cx) * 4 + quarter], 4 M + 1 entries (k_bin_cells<FINE>)
  int32_t mx, my, mzl, slab;
  FastDiv div_mx, div_my;
  float rc2;                                // largest float <= rc * rc (nl_api.hip)
  int32_t* __restrict__ count;              // [n_rows] number_of_partners
  void* __restrict__ masks;                 // [n] rows of 64 hit words, one row per sorted slot
  const void* __restrict__ key_pointer;     // [n_rows + 1] (overflow cells, FILL)
  int32_t* __restrict__ list;
  const int64_t* __restrict__ total;
  int64_t capacity;
  uint32_t* __restrict__ status;
  int32_t* __restrict__ over_list;          // local cell indices of the cells k_rows_overflow searches
  int32_t* __restrict__ over_count;         // a meta word next to the status word, zeroed with it
  int32_t wide;
  unsigned long long* dbg_buf;              // timing experiments only (-DNL_STAMP / -DNL_STAMP_FILL): per-phase wave cycles
};

// What a wave knows about the stencil of cell (cx, cy, cz).  Lane s < 36 holds the window staged as slot s: x-cells
// cx-1 .. cx+1 of one fine row, as one or (cells at the periodic wrap in x) two pieces of the sorted array.
struct RowsCtx {
  int32_t srcA, lenA, srcB, lenB;  // per lane: the window's pieces (B: the part behind the wrap in x)
  int32_t off;                     // per lane: exclusive offset of the window in the staged stream (lane 36: total_j)
  int32_t total_j;                 // particles of the whole stencil
  int32_t own_beg[4], own_n[4];    // the i-cell's four fine cells (quarters): first sorted slot, particles
  int32_t cx, cy, cz;
};

// Where fine_start keeps the first sorted slot of x-cell c (0 .. mx: mx = the end) of quarter q of the row of x-cells whose
// first cell is `base` = row * mx: the four quarters of a cell side by side, so that the 72 entries a cell's window table
// reads lie in 9-18 cache lines, not 36-72.  (The particles themselves are sorted by (row, quarter, x-cell).)
__device__ __forceinline__ int32_t rows_fine_index(int32_t base, int32_t mx, int32_t q, int32_t c) {
  if (c < mx) return (base + c) * 4 + q;
  return q < 3 ? base * 4 + q + 1 : (base + mx) * 4;  // the end of a fine row = the start of the next one
}

// Window tables of cell (cx, cy, cz; cz: local layer).  Returns false for an empty cell (or a build whose binning
// flagged the particles as inconsistent: see cell_setup_at).
__device__ __forceinline__ bool rows_windows(const RowsArgs& a, int lane, int32_t cx, int32_t cy, int32_t cz, RowsCtx& c) {
  c.cx = cx, c.cy = cy, c.cz = cz;
  const uint32_t st_word = *a.status;
  const int32_t mx = a.mx;
  const bool xwrap = cx == 0 || cx == mx - 1;  // (uniform) windows of two pieces
  // lanes 0..35: the windows; lanes 40..47: the two ends of the i-cell's four fine cells
  int32_t iA0 = 0, iA1 = 0, iB0 = 0, iB1 = 0;
  if (lane < ROWS_WIN) {
    const int32_t wz = lane / 3, dy = lane - wz * 3 - 1;
    int32_t y = cy + dy, z = cz + (wz >> 2) - 1;
    if (y < 0) y += a.my;
    if (y >= a.my) y -= a.my;
    if (!a.slab) {
      if (z < 0) z += a.mzl;
      if (z >= a.mzl) z -= a.mzl;
    }
    const int32_t base = (y + z * a.my) * mx, qz = wz & 3;
    iA0 = rows_fine_index(base, mx, qz, cx - 1), iA1 = rows_fine_index(base, mx, qz, cx + 2);
    if (xwrap) {
      if (cx == 0) iA0 = rows_fine_index(base, mx, qz, mx - 1), iA1 = rows_fine_index(base, mx, qz, mx), iB0 = rows_fine_index(base, mx, qz, 0), iB1 = rows_fine_index(base, mx, qz, 2);
      else iA0 = rows_fine_index(base, mx, qz, mx - 2), iA1 = rows_fine_index(base, mx, qz, mx), iB0 = rows_fine_index(base, mx, qz, 0), iB1 = rows_fine_index(base, mx, qz, 1);
    }
  } else if (lane >= 40 && lane < 48) {
    const int32_t j = lane - 40;
    iA0 = rows_fine_index((cy + cz * a.my) * mx, mx, j >> 1, cx + (j & 1));
    iA1 = iA0;
  }
  const int32_t vA0 = a.fine_start[iA0], vA1 = a.fine_start[iA1];
  int32_t vB0 = 0, vB1 = 0;
  if (xwrap) vB0 = a.fine_start[iB0], vB1 = a.fine_start[iB1];
  c.srcA = vA0, c.lenA = lane < ROWS_WIN ? vA1 - vA0 : 0;
  c.srcB = vB0, c.lenB = lane < ROWS_WIN ? vB1 - vB0 : 0;
  int32_t ni = 0;
#pragma unroll
  for (int f = 0; f < 4; f++) {
    c.own_beg[f] = __builtin_amdgcn_readlane(vA0, 40 + 2 * f);
    c.own_n[f] = __builtin_amdgcn_readlane(vA0, 41 + 2 * f) - c.own_beg[f];
    ni += c.own_n[f];
  }
  if (ni <= 0 || (st_word & ST_DOMAIN)) return false;
  const int32_t len = c.lenA + c.lenB;
  const int32_t incl = scan64_dpp(len);
  c.off = incl - len;
  c.total_j = __builtin_amdgcn_readlane(incl, 63);
  return true;
}

// The workgroup's cell, XCD-aware (see xcd_cell_index).
__device__ __forceinline__ void rows_cell_of_block(const RowsArgs& a, int32_t& cx, int32_t& cy, int32_t& cz) {
  const int32_t w = xcd_cell_index();
  const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx);
  cx = w - wy * a.mx;
  const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my);
  cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
}

// The i-particles of a cell, in the order quarter 0, 1, 2, 3, are dealt to the four waves of the COUNT sweep in equal
// SHARES (a workgroup keeps its LDS and wave slots until its last wave ends, and the quarters hold 9.7 +- 3.1 particles
// at BASELINE config 2: one wave per quarter left a fifth of the slots idle).  Particles of quarters qlo .. qhi searched
// together walk the quarter planes qlo .. qhi + 8: stream slots [s1, s1 + 64 ntiles), s1 = off[3 qlo] -- one piece, 27
// windows plus 3 per quarter boundary.  Bit g of lane l of a hit word of any of their rows = slot s1 + 64 g + l.
struct RowsQ {
  int32_t p1, p2, p3, n;      // particles in front of quarter 1, 2, 3; all
  int32_t ob0, ob1, ob2, ob3;  // sorted slot of particle seq of quarter q: seq + ob_q
};
__device__ __forceinline__ RowsQ rows_q(const RowsCtx& c) {
  RowsQ r;
  r.p1 = c.own_n[0], r.p2 = r.p1 + c.own_n[1], r.p3 = r.p2 + c.own_n[2], r.n = r.p3 + c.own_n[3];
  r.ob0 = c.own_beg[0], r.ob1 = c.own_beg[1] - r.p1, r.ob2 = c.own_beg[2] - r.p2, r.ob3 = c.own_beg[3] - r.p3;
  return r;
}
__device__ __forceinline__ int32_t rows_quarter_of(const RowsQ& r, int32_t seq) { return (seq >= r.p1) + (seq >= r.p2) + (seq >= r.p3); }
struct RowsShare {
  int32_t lo, hi, qlo, qhi;
};
__device__ __forceinline__ int32_t rows_first_of(const RowsQ& r, int32_t q) {  // first particle of quarter q (q = 4: the end)
  return q <= 0 ? 0 : q == 1 ? r.p1 : q == 2 ? r.p2 : q == 3 ? r.p3 : r.n;
}
__device__ __forceinline__ RowsShare rows_share(const RowsQ& r, int32_t w) {
  RowsShare sh;
#if NL_ROWS_SHARES
  sh.lo = (int32_t)(((uint32_t)r.n * (uint32_t)w) >> 2), sh.hi = (int32_t)(((uint32_t)r.n * (uint32_t)(w + 1)) >> 2);
  sh.qlo = rows_quarter_of(r, sh.lo), sh.qhi = rows_quarter_of(r, max(sh.hi - 1, sh.lo));
#else  // wave w = quarter w
  sh.lo = rows_first_of(r, w), sh.hi = rows_first_of(r, w + 1), sh.qlo = sh.qhi = w;
#endif
  return sh;
}
// A share that lies in more than two quarters (a nearly empty quarter in its middle) is searched in two PARTS, quarters
// qlo, qlo + 1 and qlo + 2, qlo + 3, so that a piece never spans more than 30 windows (16 tiles of a 16-bit hit word
// at BASELINE config 2).  Part k of share sh: particles [lo, hi), the piece [s1, s1 + 64 ntiles) they walk.
struct RowsPart {
  int32_t lo, hi, s1, ntiles;
};
__device__ __forceinline__ RowsPart rows_part(const RowsCtx& c, const RowsQ& r, const RowsShare& sh, int32_t k) {
  RowsPart pt;
  const int32_t bq = sh.qlo + 2 * k, eq = min(sh.qhi, bq + 1);
  pt.lo = max(sh.lo, rows_first_of(r, bq)), pt.hi = min(sh.hi, rows_first_of(r, bq + 2));
  pt.s1 = 0, pt.ntiles = 0;
  if (pt.hi > pt.lo) {  // (uniform)
    pt.s1 = __builtin_amdgcn_readlane(c.off, 3 * bq);
    pt.ntiles = (__builtin_amdgcn_readlane(c.off, 3 * eq + ROWS_SPAN) - pt.s1 + WAVE - 1) >> 6;
  }
  return pt;
}
// (sums of "value or 0", not selects between members of a struct: those become a table in scratch memory)
__device__ __forceinline__ int32_t rows_pick(int32_t seq, const RowsQ& r, int32_t v0, int32_t v1, int32_t v2, int32_t v3) {
  return v0 + (seq >= r.p1 ? v1 - v0 : 0) + (seq >= r.p2 ? v2 - v1 : 0) + (seq >= r.p3 ? v3 - v2 : 0);
}
// Where the particles of quarter q sit in the staged stream, minus their sequence number: in window (dz = 0, quarter q,
// dy = 0), in its A piece unless the cell is cell 0 of a wrapped row (then A is cell mx - 1 alone).
__device__ __forceinline__ int32_t rows_own_pos(const RowsCtx& c, const RowsQ& r, int32_t q, int32_t pq, int32_t ibeg) {
  const int32_t wown = (4 + q) * 3 + 1;
  const int32_t o_off = __builtin_amdgcn_readlane(c.off, wown);
  if (c.cx == 0) return o_off + __builtin_amdgcn_readlane(c.lenA, wown) + (ibeg - __builtin_amdgcn_readlane(c.srcB, wown)) - pq;
  return o_off + (ibeg - __builtin_amdgcn_readlane(c.srcA, wown)) - pq;
}

// Does the cell fit RowsCfg<V>: the stream in the LDS buffer, every share's piece in the bits of a hit word, a share in
// the lanes of a wave.  The same answer in every wave of the COUNT sweep and of the expansion (cells that do not fit are
// searched by k_rows_overflow).
template <int V> __device__ __forceinline__ bool rows_fits(const RowsCtx& c, const RowsQ& r) {
  constexpr int32_t reach = (int32_t)(8 * sizeof(typename RowsCfg<V>::word_t)) * WAVE;
  if (c.total_j > RowsCfg<V>::CAP || r.n > 4 * WAVE) return false;
  if (c.total_j <= reach) return true;
  int32_t most = 0;
#pragma unroll
  for (int w = 0; w < 4; w++) {
    const RowsShare sh = rows_share(r, w);
    most = max(most, max(rows_part(c, r, sh, 0).ntiles, rows_part(c, r, sh, 1).ntiles));
  }
  return most * WAVE <= reach;
}

// The hit words of a row in memory: 64 words, word l = lane l's.  16-bit words are stored two to a dword by the even
// lanes (the odd lane's word comes over by DPP): sub-dword vector stores are served at a fraction of the dword rate --
// with global_store_short the COUNT sweep took 0.295 ms at BASELINE config 2, 0.115 ms without its stores
// (profiles/r03_fine_rows_investigation.txt) -- and the expansion reads the dword back in both lanes.
template <typename word_t> __device__ __forceinline__ void rows_store_word(void* masks, int32_t slot, int lane, uint32_t w) {
  if constexpr (sizeof(word_t) == 2) {
    const uint32_t odd = (uint32_t)__builtin_amdgcn_update_dpp(0, (int32_t)w, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
    uint32_t* const row = static_cast<uint32_t*>(masks) + (size_t)slot * (WAVE / 2);
    if (!(lane & 1)) row[lane >> 1] = (w & 0xffffu) | (odd << 16);
  } else {
    static_cast<uint32_t*>(masks)[(size_t)slot * WAVE + lane] = w;
  }
}
template <typename word_t> __device__ __forceinline__ uint32_t rows_load_word(const void* masks, int32_t slot, int lane) {
  if constexpr (sizeof(word_t) == 2) {
    const uint32_t d = static_cast<const uint32_t*>(masks)[(size_t)slot * (WAVE / 2) + (lane >> 1)];
    return (lane & 1) ? d >> 16 : d & 0xffffu;
  } else {
    return static_cast<const uint32_t*>(masks)[(size_t)slot * WAVE + lane];
  }
}

// One group of up to GC i-particles (GC even; an odd group is padded with a particle that is never in range) against
// the tiles of the wave's piece.  The hit decision is search_group's sign-bit form (12 two-operand vector instructions
// per test, nothing scalar); positions in SGPRs, ids in VGPRs (uniform values: 48 more SGPRs would not fit 8 waves per
// SIMD).  Returns in words[k] the hit word of i-particle k (bit g = stream slot s1 + 64 g + lane accepted) and in lane
// k its count.
template <int GC, bool FULL>
__device__ __forceinline__ int32_t rows_group(const Pos<float>* tile, int lane, int32_t total_j, int32_t s1, int32_t ntiles, float rc2,
                                              const Pos<float>& pi_l, int32_t pos_l, uint32_t* words) {
  float xi[GC], yi[GC], zi[GC];
  uint32_t gi1[GC], bits[GC];
#pragma unroll
  for (int k = 0; k < GC; k++) {
    xi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.x), k));
    yi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.y), k));
    zi[k] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.z), k));
    if (!FULL) {
      gi1[k] = (uint32_t)__builtin_amdgcn_readlane(pi_l.gid, k) + 1u;
      asm volatile("" : "+v"(gi1[k]));  // kept in a vector register
    }
    bits[k] = 0;
  }
  auto test_tile = [&](const Pos<float>& pj) {
#pragma unroll
    for (int k = 0; k < GC; k++) {
      if (k && k % 3 == 0) __builtin_amdgcn_sched_barrier(0);  // (left alone the compiler hoists every subtraction to the top)
      const float dx = sub_rn(pj.x, xi[k]), dy = sub_rn(pj.y, yi[k]), dz = sub_rn(pj.z, zi[k]);
      const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
      uint32_t w = __builtin_bit_cast(uint32_t, sub_rn(rc2, r2));  // sign clear <=> !(r2 > rc2)
      if (!FULL) w |= (uint32_t)pj.gid - gi1[k];                   // sign clear <=> gid_j > gid_i
      bits[k] = __builtin_amdgcn_alignbit(bits[k], w, 31);
    }
  };
  // tile j reads slot s1 + 64 j + lane, clamped to the sentinel behind the stream (byte offsets: one add, one min)
  const uint32_t base = (uint32_t)(s1 + lane) * 16u, top = (uint32_t)total_j * 16u;
  const char* const tb = reinterpret_cast<const char*>(tile);
  auto tile_at = [&](int32_t j) { return *reinterpret_cast<const Pos<float>*>(tb + min(base + (uint32_t)j * (WAVE * 16u), top)); };
  Pos<float> pa = tile_at(0), pb;
  int32_t j = (NL_ROWS_DIAG & 4) ? ntiles : 0;
  __builtin_amdgcn_s_setprio(0);
  for (; j + 1 < ntiles; j += 2) {
    pb = tile_at(j + 1);
    test_tile(pa);
    pa = tile_at(min(j + 2, ntiles - 1));
    test_tile(pb);
  }
  if (j < ntiles) test_tile(pa);
  __builtin_amdgcn_s_setprio(3);
  uint32_t tot[GC];
#pragma unroll
  for (int k = 0; k < GC; k++) {
    uint32_t w = __brev(~bits[k]) >> (32 - ntiles);  // tile j ended at bit ntiles - 1 - j
    if (FULL) {  // the row's own particle (distance 0: in range) sits at stream slot pos_l of lane k
      const int32_t sk = __builtin_amdgcn_readlane(pos_l, k) - s1;
      if (lane == (sk & (WAVE - 1))) w &= ~(1u << (sk >> 6));
    }
    words[k] = w;
  }
#pragma unroll
  for (int k = 0; k < GC; k += 2) {
    const uint32_t two = wave_sum_dpp((uint32_t)__popc(words[k]) | (uint32_t)__popc(words[k + 1]) << 16);
    tot[k] = two & 0xffffu, tot[k + 1] = two >> 16;
  }
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < GC; k++) mine = lane == k ? tot[k] : mine;
  return (int32_t)mine;
}

// Stages the 36 windows back to back by LDS-DMA (16 bytes per lane: the particles; 4: their ids; no registers): wave v
// of NW takes slots v, v + NW, ...
__device__ __forceinline__ void rows_dma(const Pos<float>* g, Pos<float>* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ void rows_dma(const int32_t* g, int32_t* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
// One LDS-DMA instruction costs the issuing wave 450-950 cycles here whatever its lanes carry (tools/rows_phases.py),
// so the stream is staged in pieces of 64 SLOTS, not window by window: lane l of piece k fetches the particle that
// belongs in slot 64 k + l from wherever its window lies (17 instructions per cell instead of 36+).  The window of a slot:
// the last one that starts at or before the piece (a ballot over the lanes that hold the window offsets), then the few
// that start inside it, one compare + select each.  Cells at the periodic wrap in x (windows of two pieces) go window by
// window.
template <int NW, typename E>
__device__ __forceinline__ void rows_stage(const RowsCtx& c, const E* __restrict__ src, E* dst, int lane, int wave, bool xwrap) {
  if (NL_ROWS_GATHER && !xwrap) {
    const int32_t delta_l = c.srcA - c.off;  // (per lane = per window) sorted-array index minus stream slot
    const int32_t npieces = (c.total_j + WAVE - 1) >> 6;
#pragma unroll 1
    for (int32_t k = wave; k < npieces; k += NW) {
      const int32_t p0 = k * WAVE, p = p0 + lane;
      const uint64_t before = __builtin_amdgcn_ballot_w64(lane < ROWS_WIN && c.off <= p0);
      int32_t w = (int32_t)__popcll(before) - 1;
      int32_t delta = __builtin_amdgcn_readlane(delta_l, w);  // (uniform so far)
#pragma unroll 1
      for (w++; w < ROWS_WIN; w++) {
        const int32_t o = __builtin_amdgcn_readlane(c.off, w);
        if (o >= p0 + WAVE) break;
        const int32_t d = __builtin_amdgcn_readlane(delta_l, w);
        delta = p >= o ? d : delta;
      }
      if (p < c.total_j) rows_dma(src + p + delta, dst + p0);
    }
    return;
  }
#pragma unroll 1
  for (int w = wave; w < ROWS_WIN; w += NW) {
    const int32_t off = __builtin_amdgcn_readlane(c.off, w);
    const int32_t lenA = __builtin_amdgcn_readlane(c.lenA, w), srcA = __builtin_amdgcn_readlane(c.srcA, w);
#pragma unroll 1
    for (int32_t kb = 0; kb < lenA; kb += WAVE) {
      if (kb + lane < lenA) rows_dma(src + srcA + kb + lane, dst + off + kb);
    }
    const int32_t lenB = __builtin_amdgcn_readlane(c.lenB, w);
    if (lenB > 0) {
      const int32_t srcB = __builtin_amdgcn_readlane(c.srcB, w);
#pragma unroll 1
      for (int32_t kb = 0; kb < lenB; kb += WAVE) {
        if (kb + lane < lenB) rows_dma(src + srcB + kb + lane, dst + off + lenA + kb);
      }
    }
  }
}

template <int V, bool FULL>
__global__ void __launch_bounds__(ROWS_WAVES* WAVE, RowsCfg<V>::WS) __attribute__((amdgpu_num_sgpr(96))) k_sweep_rows_f32(RowsArgs a) {
  typedef RowsCfg<V> Cfg;
  typedef typename Cfg::word_t word_t;
  constexpr int CAP = Cfg::CAP;
  __shared__ __attribute__((aligned(32))) Pos<float> tile[CAP + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(3);  // everything but the tile loop (see cell_search: the arbiter otherwise starves the set-up)
#if NL_STAMP
  uint64_t t_prev = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0}, n_tests = 0;  // wave-uniform: SGPRs
  auto stamp = [&](int phase) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    t_acc[phase] += now - t_prev;
    t_prev = now;
  };
#endif
  if (NL_ROWS_EXIT == 1) return;
  int32_t cx, cy, cz;
  rows_cell_of_block(a, cx, cy, cz);
  RowsCtx c;
  if (!rows_windows(a, lane, cx, cy, cz, c)) return;
  const RowsQ rq = rows_q(c);
  if (!rows_fits<V>(c, rq)) {  // (uniform over the workgroup) searched straight from memory by k_rows_overflow
    if (tid == 0) a.over_list[atomicAdd(a.over_count, 1)] = cx + (cy + cz * a.my) * a.mx;
    return;
  }
#if NL_STAMP
  stamp(0);  // cell, window table (loads), scan
#endif
  if (NL_ROWS_EXIT == 2) {
    if (c.total_j == 0x7fffffff) a.count[lane] = c.off;
    return;
  }
  rows_stage<ROWS_WAVES>(c, a.sorted, tile, lane, wave, cx == 0 || cx == a.mx - 1);
  if (NL_ROWS_EXIT == 3) return;
  if (tid == 0) {  // far outside any box (finite r2, never in range) and an id that is never the upper one
    Pos<float> sentinel;
    sentinel.x = 1.0e18f, sentinel.y = 0.f, sentinel.z = 0.f, sentinel.gid = INT32_MIN;
    tile[c.total_j] = sentinel;
  }
  const RowsShare sh = rows_share(rq, wave);
  // stream slot of particle seq: seq + op_q (q: its quarter)
  const int32_t op0 = rows_own_pos(c, rq, 0, 0, c.own_beg[0]), op1 = rows_own_pos(c, rq, 1, rq.p1, c.own_beg[1]);
  const int32_t op2 = rows_own_pos(c, rq, 2, rq.p2, c.own_beg[2]), op3 = rows_own_pos(c, rq, 3, rq.p3, c.own_beg[3]);
#if NL_STAMP
  stamp(1);  // DMA issued
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if NL_STAMP
  stamp(2);  // stream landed, barrier
#endif
  if (NL_ROWS_EXIT == 4) {
    if (tile[lane].gid == 0x7ffffff1) a.count[lane] = 1;
    return;
  }
  if (sh.hi <= sh.lo) return;
#pragma unroll 1
  for (int32_t part = 0; part < 2; part++) {
    const RowsPart pt = rows_part(c, rq, sh, part);
    const int32_t n = pt.hi - pt.lo;
    if (n <= 0) break;  // (a second part only where the share lies in more than two quarters)
    const int32_t ngroups = (n + ROWS_G - 1) / ROWS_G;
    const int32_t gbase = n / ngroups, grem = n - gbase * ngroups;  // group g: gbase (+ 1 if g < grem) particles
#pragma unroll 1
    for (int32_t g = 0; g < ngroups; g++) {
      const int32_t i0 = g * gbase + min(g, grem);
      const int32_t gcount = gbase + (g < grem ? 1 : 0);
      const int32_t seq = pt.lo + i0 + min(lane, gcount - 1);  // lane k: particle k of the group
      const int32_t slot_l = seq + rows_pick(seq, rq, rq.ob0, rq.ob1, rq.ob2, rq.ob3);
      const int32_t pos_l = seq + rows_pick(seq, rq, op0, op1, op2, op3);
      Pos<float> pi_l = tile[pos_l];
      const int32_t row_l = a.sorted_row[slot_l];
      if (lane >= gcount) pi_l.x = 1.0e18f;  // the padding of an odd group: never in range
      uint32_t words[12];
      int32_t mine;
#if NL_STAMP
      stamp(3);  // group set-up
      n_tests += (uint64_t)(gcount * pt.ntiles);
#endif
      switch ((gcount + 1) >> 1) {
        case 1: mine = rows_group<2, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
        case 2: mine = rows_group<4, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
        case 3: mine = rows_group<6, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
        case 4: mine = rows_group<8, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
        case 5: mine = rows_group<10, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
        default: mine = rows_group<12, FULL>(tile, lane, c.total_j, pt.s1, pt.ntiles, a.rc2, pi_l, pos_l, words); break;
      }
#if NL_STAMP
      stamp(4);  // rows_group: readlanes, tile loop, words and counts
#endif
      if (lane < gcount && (!(NL_ROWS_DIAG & 2) || a.wide == 7)) a.count[row_l] = mine;
#pragma unroll
      for (int k = 0; k < ROWS_G; k++) {
        if (k < gcount && (!(NL_ROWS_DIAG & 1) || a.wide == 7)) rows_store_word<word_t>(a.masks, __builtin_amdgcn_readlane(slot_l, k), lane, words[k]);
      }
#if NL_STAMP
      stamp(5);  // stores
#endif
    }
  }
#if NL_STAMP
  if (lane == 0) {  // one set of atomics per wave, spread over 1024 slots (the host sums them)
    unsigned long long* const slot = a.dbg_buf + 64 + (blockIdx.x & 1023) * 16;
    for (int ph = 0; ph < 6; ph++) atomicAdd(slot + ph, (unsigned long long)t_acc[ph]);
    atomicAdd(slot + 8, (unsigned long long)n_tests);
    atomicAdd(slot + 9, 1ull);
  }
#endif
}

// ------------------------------------------------------------------------------------------ the list from the hit words
// Two waves per cell, each expands half of the cell's rows.  The expansion is bound by the rate at which a CU ISSUES
// vector-memory instructions (7-26 ns each whatever they carry, tools/dma_rate.hip; a wave of the first version issued
// 78 of them, 24 000 of its 58 000 cycles), so everything moves in the widest pieces there are:
//   * the ids of the cell's stream: LDS-DMA (dword) from the compact id array, one instruction per 64 stream slots
//     (rows_stage), so bit g of lane l of a row names ids[s1 + 64 g + l], s1 the start of the piece its group walked in
//     the COUNT sweep;
//   * the hit words of the wave's rows: LDS-DMA, 16 bytes per lane = eight 128-byte rows per instruction, then one LDS
//     read per row and lane; all of them before the wave's first store (vmcnt retires in order: a load issued behind
//     stores waits for every one of them);
//   * the rows: four at a time, assembled in LDS as in k_fill_masks (a DPP prefix sum of the popcounts of a row's 64
//     words, every lane takes its set bits one by one), then written out back to back -- lane t of a store carries entry
//     t of the four rows' concatenation, so a store is full whatever the rows' lengths (5 stores per four rows of 73
//     entries instead of 8).
constexpr int ROWS_FW = 2;          // waves per workgroup of the expansion (16 workgroups per CU)
constexpr int ROWS_RMAX = 96;       // longest row assembled in LDS (longer rows are written entry by entry)
constexpr int ROWS_RBYTES = 2560;   // LDS per wave: first the hit words of its rows, then the four rows being assembled

// Row r of a wave (lane r holds its data): particle lo + r of the sequence quarter 0, 1, 2, 3 -> its sorted slot, the
// stream slot that bit 0 of lane 0 of its hit words stands for, its offset in the list.  (A struct of scalars with a
// forced-inline member, not a lambda: a closure that refers to the kernel argument makes the compiler keep the argument
// block in scratch memory.)
template <typename OFF, typename word_t, int RB> struct FillRows {
  int32_t lo, n;
  RowsQ rq;
  int32_t l1, l2, l3, ql0, ql1, ql2, ql3;  // first particle of share 1, 2, 3 of the COUNT sweep; first quarter of the four shares
  const char* masks;
  const OFF* key_pointer;
  const int32_t* sorted_row;
  // issues the loads of rows r0 .. r0 + RB - 1: list offsets into base_l, hit words by LDS-DMA into `rows`
  __device__ __forceinline__ void issue(int32_t r0, int lane, int32_t off_l, int32_t& s1_l, OFF& base_l, char* rows) const {
    constexpr int ROWB = WAVE * (int)sizeof(word_t), LPR = ROWB / 16, RPP = WAVE / LPR;  // bytes per row, lanes / rows per DMA piece
    const int32_t seq = lo + min(r0 + lane, max(n - 1, 0));  // (a wave without rows reads one valid row and drops it)
    const int32_t slot_l = seq + rows_pick(seq, rq, rq.ob0, rq.ob1, rq.ob2, rq.ob3);
    // the piece the row's group walked in the COUNT sweep starts at the first quarter of its share, or two quarters on
    // (rows_part)
    const int32_t qlo = ql0 + (seq >= l1 ? ql1 - ql0 : 0) + (seq >= l2 ? ql2 - ql1 : 0) + (seq >= l3 ? ql3 - ql2 : 0);
    const int32_t bq = rows_quarter_of(rq, seq) - qlo >= 2 ? qlo + 2 : qlo;
    s1_l = __shfl(off_l, 3 * bq, WAVE);
    base_l = key_pointer[sorted_row[slot_l]];  // (two dependent loads per batch, in the shadow of the DMA below: no gather kernel in front)
    const int32_t nb = min(n - r0, RB);
#pragma unroll
    for (int pc = 0; pc < (RB + RPP - 1) / RPP; pc++) {
      if (pc * RPP < nb) {  // (uniform)
        const int32_t r = pc * RPP + lane / LPR;  // this lane's row of the batch and its 16 bytes of it
        const int32_t slot = __shfl(slot_l, min(r, RB - 1), WAVE);
        if (r < nb)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(masks + (size_t)slot * ROWB + (lane % LPR) * 16),
                                           (__attribute__((address_space(3))) void*)(rows + pc * 1024), 16, 0, 0);
      }
    }
  }
};

template <int V, bool FULL, typename OFF>
__global__ void __launch_bounds__(ROWS_FW* WAVE, (sizeof(OFF) == 8 && RowsCfg<V>::WF > 6 ? 6 : RowsCfg<V>::WF)) __attribute__((amdgpu_num_sgpr(96)))
k_fill_rows(RowsArgs a) {
  typedef RowsCfg<V> Cfg;
  typedef typename Cfg::word_t word_t;
  constexpr int CAP = Cfg::CAP, ROWB = WAVE * (int)sizeof(word_t), RB = ROWS_RBYTES / ROWB / 4 * 4;  // 20 rows of 16-bit words, 8 of 32-bit words
  static_assert(4 * ROWS_RMAX * 4 <= ROWS_RBYTES && RB % 4 == 0, "the assembly buffer reuses the rows' LDS");
  __shared__ __attribute__((aligned(32))) int32_t lds[(CAP + 1 + 3) / 4 * 4 + ROWS_FW * ROWS_RBYTES / 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if NL_STAMP_FILL
  uint64_t ft_prev = __builtin_amdgcn_s_memtime(), ft_acc[6] = {0, 0, 0, 0, 0, 0};
  auto fstamp = [&](int phase) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    ft_acc[phase] += now - ft_prev;
    ft_prev = now;
  };
#endif
  const int64_t total = a.total[0];
  int32_t cx, cy, cz;
  rows_cell_of_block(a, cx, cy, cz);
  RowsCtx c;
  const bool ok = rows_windows(a, lane, cx, cy, cz, c);
#if NL_STAMP_FILL
  fstamp(0);  // cell, window table
#endif
  if (total > a.capacity) {  // the list is too small: the host grows it and runs the expansion again
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
    return;
  }
  if (!ok) return;
  const RowsQ rq = rows_q(c);
  if (!rows_fits<V>(c, rq)) return;  // k_rows_overflow (the COUNT sweep listed the cell)
  char* const rows = reinterpret_cast<char*>(lds + (CAP + 1 + 3) / 4 * 4) + wave * ROWS_RBYTES;
  int32_t* const cw = reinterpret_cast<int32_t*>(rows);
  FillRows<OFF, word_t, RB> fr;
  fr.rq = rq;
  fr.lo = (int32_t)(((uint32_t)rq.n * (uint32_t)wave) / ROWS_FW);
  fr.n = (int32_t)(((uint32_t)rq.n * (uint32_t)(wave + 1)) / ROWS_FW) - fr.lo;
  {
    const RowsShare s0 = rows_share(rq, 0), s1 = rows_share(rq, 1), s2 = rows_share(rq, 2), s3 = rows_share(rq, 3);
    fr.l1 = s1.lo, fr.l2 = s2.lo, fr.l3 = s3.lo;
    fr.ql0 = s0.qlo, fr.ql1 = s1.qlo, fr.ql2 = s2.qlo, fr.ql3 = s3.qlo;
  }
  fr.masks = static_cast<const char*>(a.masks), fr.key_pointer = static_cast<const OFF*>(a.key_pointer), fr.sorted_row = a.sorted_row;
  const int32_t n = fr.n;  // rows of this wave
  int32_t s1_l = 0;
  OFF base_l = 0;
  int32_t* const list = a.list;
  fr.issue(0, lane, c.off, s1_l, base_l, rows);
  rows_stage<ROWS_FW>(c, a.sorted_gid, lds, lane, wave, cx == 0 || cx == a.mx - 1);
#if NL_STAMP_FILL
  fstamp(1);  // row loads and id DMA issued
#endif
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // ids staged (and this wave's words here)
#if NL_STAMP_FILL
  fstamp(2);  // landed, barrier
#endif
  const char* const idb = reinterpret_cast<const char*>(lds) + lane * 4;
#pragma unroll 1
  for (int32_t r0 = 0; r0 < n; r0 += RB) {
    if (r0) {  // (a wave with more than RB rows)
      fr.issue(r0, lane, c.off, s1_l, base_l, rows);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    uint32_t w[RB];  // the hit words of the batch: out of the LDS rows, which then serve as the assembly buffer
#pragma unroll
    for (int u = 0; u < RB; u++) w[u] = (uint32_t) * reinterpret_cast<const word_t*>(rows + u * ROWB + lane * sizeof(word_t));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u0 = 0; u0 < RB; u0 += 4) {
      if (r0 + u0 >= n) continue;  // wave-uniform
      uint32_t word[4], ptr[4], gofs[4];
      int32_t nrow[4];
      OFF base[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        word[p] = r0 + u0 + p < n ? w[u0 + p] : 0u;
        const int32_t cnt = __popc(word[p]);
        const int32_t incl = scan64_dpp(cnt);
        nrow[p] = __builtin_amdgcn_readlane(incl, 63);
        ptr[p] = (uint32_t)(incl - cnt);  // place inside the row
        gofs[p] = (uint32_t)__builtin_amdgcn_readlane(s1_l, u0 + p) * 4u;  // the ids of the row's piece start there
        if constexpr (sizeof(OFF) == 8) {
          const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)base_l, u0 + p);
          const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)((uint64_t)base_l >> 32), u0 + p);
          base[p] = (OFF)(((uint64_t)bhi << 32) | blo);
        } else {
          base[p] = (OFF)__builtin_amdgcn_readlane((int32_t)base_l, u0 + p);
        }
      }
      const int32_t nmax = max(max(nrow[0], nrow[1]), max(nrow[2], nrow[3]));
#if NL_STAMP_FILL
      fstamp(3);  // words arrived, popcounts, scans
#endif
      if (nmax <= ROWS_RMAX) {
#pragma unroll
        for (int p = 0; p < 4; p++) ptr[p] += p * ROWS_RMAX;
        while (word[0] | word[1] | word[2] | word[3]) {
          int32_t val[4];
          bool on[4];
#pragma unroll
          for (int p = 0; p < 4; p++) {
            on[p] = word[p] != 0;
            const uint32_t t = on[p] ? (uint32_t)__ffs(word[p]) - 1u : 0u;
            val[p] = *reinterpret_cast<const int32_t*>(idb + gofs[p] + t * (WAVE * 4u));  // unconditional read of a valid slot
          }
#pragma unroll
          for (int p = 0; p < 4; p++) {
            if (on[p]) {
              cw[ptr[p]] = val[p];
              ptr[p]++;
              word[p] &= word[p] - 1;
            }
          }
        }
#if NL_STAMP_FILL
        fstamp(4);  // bit loop
#endif
        __builtin_amdgcn_wave_barrier();  // the buffer is private to the wave: LDS executes its accesses in order
        // the four rows back to back: lane t of a store carries entry t of their concatenation
        const int32_t c0 = nrow[0], c1 = c0 + nrow[1], c2 = c1 + nrow[2], c3 = c2 + nrow[3];
        for (int32_t t = lane; t - lane < c3; t += WAVE) {
          const int32_t e = t - (t >= c0 ? c0 : 0) - (t >= c1 ? c1 - c0 : 0) - (t >= c2 ? c2 - c1 : 0);
          const int32_t at = e + (t >= c0 ? ROWS_RMAX : 0) + (t >= c1 ? ROWS_RMAX : 0) + (t >= c2 ? ROWS_RMAX : 0);
          const int32_t val = cw[min(at, 4 * ROWS_RMAX - 1)];
          const OFF b = base[0] + (t >= c0 ? base[1] - base[0] : 0) + (t >= c1 ? base[2] - base[1] : 0) + (t >= c2 ? base[3] - base[2] : 0);
          if (t < c3) list[(int64_t)b + e] = val;
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
        for (int p = 0; p < 4; p++) {
          on[p] = word[p] != 0;
          const uint32_t t = on[p] ? (uint32_t)__ffs(word[p]) - 1u : 0u;
          val[p] = *reinterpret_cast<const int32_t*>(idb + gofs[p] + t * (WAVE * 4u));
        }
#pragma unroll
        for (int p = 0; p < 4; p++) {
          if (on[p]) {
            (list + (int64_t)base[p])[ptr[p]] = val[p];
            ptr[p]++;
            word[p] &= word[p] - 1;
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

// ------------------------------------------------------------------------------------------ cells that do not fit
// The cells the COUNT sweep listed (a stream beyond the LDS buffer: a cluster among ordinary cells): both passes straight
// from memory, one wave per quarter: the 64 candidates of a piece of a window in registers, the quarter's particles one
// after the other against them (lane i keeps the count of particle i).  The reach of the quarter alone: 27 windows.
template <int MODE, bool FULL, typename OFF>
__global__ void __launch_bounds__(ROWS_WAVES* WAVE) k_rows_overflow(RowsArgs a) {
  if (MODE == MODE_FILL && a.total[0] > a.capacity) return;  // (k_fill_rows has raised ST_CAPACITY)
  const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t count = *a.over_count;
  for (int32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
    const int32_t cell = a.over_list[idx];
    const int32_t row = cell / a.mx, cx = cell - row * a.mx, cz = row / a.my, cy = row - cz * a.my;
    RowsCtx c;
    if (!rows_windows(a, lane, cx, cy, cz, c)) continue;
    const int32_t ni = c.own_n[0] * (q == 0) + c.own_n[1] * (q == 1) + c.own_n[2] * (q == 2) + c.own_n[3] * (q == 3);
    const int32_t ibeg = c.own_beg[0] * (q == 0) + c.own_beg[1] * (q == 1) + c.own_beg[2] * (q == 2) + c.own_beg[3] * (q == 3);
    for (int32_t i0 = 0; i0 < ni; i0 += WAVE) {  // 64 particles of the quarter at a time: lane i holds particle i0 + i
      const int32_t nb = min(ni - i0, WAVE);
      const Pos<float> pi_l = a.sorted[ibeg + i0 + min(lane, nb - 1)];
      const int32_t row_l = a.sorted_row[ibeg + i0 + min(lane, nb - 1)];
      int64_t base_l = 0;
      if (MODE == MODE_FILL) base_l = (int64_t)static_cast<const OFF*>(a.key_pointer)[row_l];
      int32_t cnt_l = 0;
      for (int32_t wdw = 3 * q; wdw < 3 * q + ROWS_SPAN; wdw++) {
        for (int part = 0; part < 2; part++) {
          const int32_t len = __builtin_amdgcn_readlane(part ? c.lenB : c.lenA, wdw);
          const int32_t src = __builtin_amdgcn_readlane(part ? c.srcB : c.srcA, wdw);
          for (int32_t kb = 0; kb < len; kb += WAVE) {
            const bool valid = kb + lane < len;
            const Pos<float> pj = a.sorted[src + min(kb + lane, len - 1)];
            for (int32_t i = 0; i < nb; i++) {
              const float xi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.x), i));
              const float yi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.y), i));
              const float zi = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int32_t, pi_l.z), i));
              const int32_t gi = __builtin_amdgcn_readlane(pi_l.gid, i);
              const float dx = sub_rn(pj.x, xi), dy = sub_rn(pj.y, yi), dz = sub_rn(pj.z, zi);
              const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
              const bool hit = valid && !(r2 > a.rc2) && (FULL ? pj.gid != gi : pj.gid > gi);
              const uint64_t m = __builtin_amdgcn_ballot_w64(hit);
              if (MODE == MODE_FILL && hit) {
                const int32_t at = __builtin_amdgcn_readlane(cnt_l, i);
                const uint32_t blo = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)base_l, i);
                const uint32_t bhi = (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)((uint64_t)base_l >> 32), i);
                int32_t* const rowp = a.list + (int64_t)(((uint64_t)bhi << 32) | blo);
                rowp[at + (int32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = pj.gid;
              }
              if (lane == i) cnt_l += (int32_t)__popcll(m);
            }
          }
        }
      }
      if (MODE == MODE_COUNT && lane < nb) a.count[row_l] = cnt_l;
    }
  }
}

}  // namespace nl
