// nl_sweep_mfma.hpp -- a8 (pair search) for fp32 on the matrix cores.  Included by nl_kernels.hpp.
//
// The distance test of the reference (neighlist_cpu.hpp:219-223, neighlist_gpu.hpp:92-97) is
//     r2 = (dx*dx + dy*dy) + dz*dz  (fp32, every operation rounded),  pair kept unless r2 > rc2.
// The VALU sweep (search_group) pays 10 vector instructions per 64 tests for it and is issue-bound.  Here the
// 16 x 16 block of r2 - rc2 values between 16 i-particles and 16 j-particles comes out of ONE
// v_mfma_f32_16x16x4_f32:  with u = position - centre of the i-cell,
//     r2 - rc2 = (|ui|^2 - rc2)  +  [-2ux_i, -2uy_i, -2uz_i, 1] . [ux_j, uy_j, uz_j, |uj|^2]
//                 C operand            A row (K = 4)                  B column
// That value is NOT the reference's r2 bit for bit (different association, local coordinates), so it is used only
// where it is decisive: |acc| >= delta, with delta a rigorous bound on the difference (host: mfma_delta()).  The
// few elements inside the band (about 1e-4 of the accepted pairs) are re-tested with the reference's exact
// expression from the original coordinates before their sign is used, so the accepted set is exactly the reference's.
//
// Per accumulator register (64 tests) the vector work is 3.5 instructions: half a v_min3_f32 (running minimum of
// |acc|, compared with delta once per tile), gid_i - gid_j (sign = "j is the upper index"), AND with the
// accumulator (sign = accepted), and one v_alignbit that shifts the sign bit into the lane's hit word.  No scalar bookkeeping per test: the counts are
// popcounts of the hit words at the end.
//
// Work split: a workgroup of 6 waves owns an i-cell.  The stencil stream is staged ONCE into LDS as four component
// arrays (ux, uy, uz, |u|^2) + ids (26 KB: five workgroups = 30 waves per CU).  A unit of work is one i-block (16
// rows) against one half of the j-tiles (16 particles each): tiles [0, 32) or [32, ntiles); a cell of 33..48 rows has
// six units, one per wave.  Little per-wave state (one MFMA accumulator pair, 4 row constants, 4 hit words).
// Hit words (layout MASK_TILE16, 48 per row): word (t / 32)*16 + lam, bit 31 - t % 32 for tile t and lam = j % 16;
// the halves meet at a word boundary, so every word has one writer and goes straight to memory in 64-byte pieces.
// k_fill_masks<T, MASK_TILE16> expands them.
#pragma once

namespace nl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_CAP = SweepCfg<float>::CAP;  // staged particles per cell: the same single-batch limit as the masks
constexpr int MF_CSTR = MF_CAP + 16;          // component array stride: the four bases fall on banks 0, 16, 32, 48
constexpr int MF_TILE = 16;
constexpr int MF_WAVES = 6;                   // waves per workgroup
constexpr int MF_ROWS = 64;                   // most rows a cell may have on this path (k_fill_masks knows this too)
constexpr int MF_WORDS = (MF_CAP / MF_TILE + 31) / 32 * 16;  // hit words per row: 3 groups of 32 tiles x 16 lanes = 48

struct MfmaLds {
  float comp[4 * MF_CSTR];  // ux | uy | uz | |u|^2 of the staged stream
  int32_t gid[MF_CAP];
  int32_t cnt[MF_ROWS];
};
static_assert(sizeof(float) * 4 * MF_CSTR >= sizeof(Pos<float>) * SweepCfg<float>::CAP, "fallback tile fits");
static_assert(MF_WORDS == MASK16_WORDS && MF_WORDS == 48 && MF_ROWS == MASK16_MAX_ROWS, "layout shared with k_fill_masks");

// stream position -> index in the sorted array (walks the 18-entry segment table held one entry per lane)
__device__ __forceinline__ int32_t mf_stream_to_sorted(const CellCtx& c, int32_t p) {
  int32_t idx = 0;
  for (int s = 0; s < NSEG; s++) {
    const int32_t off = __builtin_amdgcn_readlane(c.seg_off, s), len = __builtin_amdgcn_readlane(c.seg_len, s),
                  src = __builtin_amdgcn_readlane(c.seg_src, s);
    if (p >= off && p < off + len) idx = src + (p - off);
  }
  return idx;
}

#ifdef MF_ABL_NO_MFMA  // ablation: the vector work on an accumulator that no MFMA produces (bv keeps the loads alive)
#define MF_MFMA(A_, B_, C_) ((C_) + (B_))
#else
#define MF_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, C_, 0, 0, 0)
#endif

// One unit: the i-block of 16 rows starting at row i0 of the cell against half `half` of the ntiles staged tiles.
// i_off: stream position of the cell's own first particle (the i-particles are part of their own stencil, so their
// local coordinates and ids are already in LDS).
__device__ __forceinline__ void mf_unit(const SweepArgs<float>& a, const CellCtx& c, MfmaLds& L, int lane,
                                        int32_t i_off, int32_t i0, int32_t half, int32_t ntiles) {
  const int kq = lane >> 4, lam = lane & 15;
  float A;
  f32x4 C;
  int32_t gi[4];
  {
    const int32_t irow = i0 + lam;
    const float u = L.comp[min(kq, 2) * MF_CSTR + i_off + min(irow, c.ni - 1)];
    A = irow < c.ni ? (kq == 3 ? 1.0f : mul_rn(-2.0f, u)) : 0.0f;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int32_t irow = i0 + 4 * kq + r;
    const int32_t p = i_off + min(irow, c.ni - 1);
    C[r] = irow < c.ni ? sub_rn(L.comp[3 * MF_CSTR + p], a.rc2) : 1.0e30f;  // padding rows: never accepted
    gi[r] = L.gid[p];
  }

  const float* const bp = L.comp + kq * MF_CSTR + lam;
  const int32_t* const gp = L.gid + lam;
  const float delta = a.delta;
  uint32_t bits[4];

  // One tile's accumulators -> one more bit in every hit word.
  auto process = [&](const f32x4& acc, int32_t gj, int32_t t) {
    // smallest |r2 - rc2| of the tile in this lane: v_min3_f32 with |.| source modifiers, half an instruction per value
    float m = __builtin_huge_valf();
#pragma unroll
    for (int r = 0; r < 4; r++) m = __builtin_fminf(m, __builtin_fabsf(acc[r]));
    // sign(acc) = accepted by distance, sign(gid_i - gid_j) = j is the upper index (ids are >= 0)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float av = acc[r];  // (bit_cast straight from the vector element picks element 0)
      const uint32_t h = __float_as_uint(av) & (uint32_t)(gi[r] - gj);
      bits[r] = __builtin_amdgcn_alignbit(bits[r], h, 31);  // (bits << 1) | (h >> 31)
    }
#ifdef MF_ABL_NO_UNC  // tools/mfma_bench ablation: no band check at all (timing only, results wrong)
    if (false) {
#else
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m < delta) != 0, 0)) {
#endif
      // rare (about 1 % of the tiles): some |r2 - rc2| is inside the error band of the matrix-core expression.
      // Re-test those elements with the reference's expression on the original coordinates and flip the bit just
      // written where the exact answer differs.  (Nothing here writes the accumulators: the common path keeps them
      // where the MFMA left them, without copies.)
      const int32_t pj_pos = t * MF_TILE + lam;
      const bool jok = pj_pos < c.total_j;
      const Pos<float> pj = a.sorted[jok ? mf_stream_to_sorted(c, pj_pos) : c.ibeg];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int32_t irow = i0 + 4 * kq + r;
        const float av = acc[r];
        if (__builtin_fabsf(av) < delta && jok && irow < c.ni) {
          const Pos<float> pi = a.sorted[c.ibeg + irow];
          const float dx = sub_rn(pj.x, pi.x), dy = sub_rn(pj.y, pi.y), dz = sub_rn(pj.z, pi.z);
          const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
          const bool want = !(r2 > a.rc2), have = (__float_as_uint(av) >> 31) != 0;
          if (want != have && gi[r] < gj) bits[r] ^= 1u;
          if (a.dbg & 8) atomicAdd(a.dbg_buf + 0, 1ull);
        }
      }
    }
  };

  // Word group g = tiles [32 g, 32 g + n): search them into `bits` (nothing to do for n <= 0), store the group's
  // words of the block's rows, add their popcounts to `total`.
  int32_t total[4] = {0, 0, 0, 0};
  auto run = [&](int32_t g, int32_t n) {
#pragma unroll
    for (int r = 0; r < 4; r++) bits[r] = 0;
    if (n > 0) {
      const int32_t tb = 32 * g, t_last = tb + n - 1;
      // Two accumulators in ping-pong: the MFMA of the next tile is issued before the vector work of the current
      // one.  Tile indices beyond the run are clamped (the result of such an MFMA is never consumed).
#ifdef MF_ABL_NO_LDS  // ablation: every step reads the same tile (the loads are hoisted out of the loop)
      auto tix = [&](int32_t) { return tb * MF_TILE; };
#else
      auto tix = [&](int32_t t) { return min(t, t_last) * MF_TILE; };
#endif
      float bv0 = bp[tix(tb)], bv1 = bp[tix(tb + 1)];
      int32_t g0 = gp[tix(tb)], g1 = gp[tix(tb + 1)];
      f32x4 acc0 = MF_MFMA(A, bv0, C), acc1;
      for (int32_t t = tb; t <= t_last; t += 2) {
        acc1 = MF_MFMA(A, bv1, C);
        bv0 = bp[tix(t + 2)];
        const int32_t g0n = gp[tix(t + 2)];
        process(acc0, g0, t);
        acc0 = MF_MFMA(A, bv0, C);
        bv1 = bp[tix(t + 3)];
        const int32_t g1n = gp[tix(t + 3)];
        if (t + 1 <= t_last) process(acc1, g1, t + 1);
        g0 = g0n, g1 = g1n;
      }
      const uint32_t up = 32u - (uint32_t)n;  // first tile of the group -> bit 31
#pragma unroll
      for (int r = 0; r < 4; r++) bits[r] <<= up;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int32_t irow = i0 + 4 * kq + r;
#ifndef MF_ABL_NO_STORE
      if (irow < c.ni) a.masks[(size_t)(c.ibeg + irow) * MF_WORDS + g * 16 + lam] = bits[r];
#endif
      total[r] += __popc(bits[r]);
    }
  };
  // half 0: word group 0; half 1: word groups 1 and 2 (one instance of the loop body for all three)
  for (int32_t g = half; g < 1 + 2 * half; g++) run(g, a.dbg & 1 ? 0 : min(32, ntiles - 32 * g));
#pragma unroll
  for (int r = 0; r < 4; r++) {
    int32_t v = total[r];
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1 within the 16 lanes of a row group
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    if (lam == 15 && v) atomicAdd(&L.cnt[i0 + 4 * kq + r], v);
  }
}

// diagnostics (NL_DEBUG_FLAGS & 4): thread 0 records the cycles since the previous stamp in the workgroup's
// record dbg_buf[64 + (block % 2048) * 8 + phase] (plain stores: later workgroups overwrite earlier ones)
__device__ __forceinline__ void mf_stamp(const SweepArgs<float>& a, int tid, int phase, unsigned long long& t_prev) {
  if (a.dbg & 4) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    if (tid == 0) a.dbg_buf[64 + (blockIdx.x & 2047) * 8 + phase] = t - t_prev;
    t_prev = t;
  }
}

__global__ void __launch_bounds__(MF_WAVES* WAVE, 8) k_sweep_mfma_f32(SweepArgs<float> a) {
  __shared__ __attribute__((aligned(16))) MfmaLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long t_prev = 0;
  if (a.dbg & 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  CellCtx c;
  if (!cell_setup(a, lane, c)) return;
  mf_stamp(a, tid, 0, t_prev);  // cell table
  if (c.total_j > MF_CAP || c.ni > MF_ROWS) {
    // stencil larger than one LDS batch (or a very full cell): counts only, by the VALU search; k_fill_masks
    // searches such cells again
    cell_search<float, MODE_COUNT, SweepCfg<float>::CAP, MF_WAVES>(a, c, reinterpret_cast<Pos<float>*>(L.comp), tid, lane,
                                                                   wave);
    return;
  }
  // centre of the i-cell: the origin of the local coordinates (any point near the cell would do)
  const float ccx = mul_rn((float)c.cx + 0.5f, a.ms[0]), ccy = mul_rn((float)c.cy + 0.5f, a.ms[1]),
              ccz = mul_rn((float)(c.cz + a.z_origin) + 0.5f, a.ms[2]);
  // where the cell's own particles sit in the stream: in segment 4 (dz = dy = 0, first x-part) or, for the cells
  // at the low x face, at the head of its wrapped part, segment 13
  int32_t i_off;
  {
    const int32_t s4 = __builtin_amdgcn_readlane(c.seg_src, 4), l4 = __builtin_amdgcn_readlane(c.seg_len, 4);
    const bool in4 = c.ibeg >= s4 && c.ibeg < s4 + l4;
    i_off = in4 ? __builtin_amdgcn_readlane(c.seg_off, 4) + (c.ibeg - s4)
                : __builtin_amdgcn_readlane(c.seg_off, 13) + (c.ibeg - __builtin_amdgcn_readlane(c.seg_src, 13));
  }

  // ---- stage the stream as component arrays.  Wave w copies segments w, w + 6, w + 12; the loads of the first
  // 192 particles of each are all issued before the first LDS write (one memory round trip per cell, not one
  // per 128 particles); longer segments (dense cells) finish in a plain loop.
  auto put = [&](int32_t p, const Pos<float>& v) {
    const float ux = sub_rn(v.x, ccx), uy = sub_rn(v.y, ccy), uz = sub_rn(v.z, ccz);
    L.comp[p] = ux, L.comp[MF_CSTR + p] = uy, L.comp[2 * MF_CSTR + p] = uz;
    L.comp[3 * MF_CSTR + p] = add_rn(add_rn(mul_rn(ux, ux), mul_rn(uy, uy)), mul_rn(uz, uz));
    L.gid[p] = v.gid;
  };
  if (!(a.dbg & 2)) {
    constexpr int NS = (NSEG + MF_WAVES - 1) / MF_WAVES, NK = 3;
    int32_t len[NS], src[NS], off[NS];
    Pos<float> v[NS][NK];
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const int sg = wave + q * MF_WAVES;  // wave-uniform; < 32, lanes >= NSEG hold empty segments
      len[q] = __builtin_amdgcn_readlane(c.seg_len, sg);
      src[q] = __builtin_amdgcn_readlane(c.seg_src, sg);
      off[q] = __builtin_amdgcn_readlane(c.seg_off, sg);
#pragma unroll
      for (int k = 0; k < NK; k++)  // (unconditional: an empty segment re-reads one valid particle)
        v[q][k] = a.sorted[src[q] + max(min(k * WAVE + lane, len[q] - 1), 0)];
    }
#pragma unroll
    for (int q = 0; q < NS; q++) {
#pragma unroll
      for (int k = 0; k < NK; k++)
        if (k * WAVE + lane < len[q]) put(off[q] + k * WAVE + lane, v[q][k]);
      for (int32_t k = NK * WAVE + lane; k < len[q]; k += WAVE) put(off[q] + k, a.sorted[src[q] + k]);
    }
  }
  {  // sentinels up to the next tile boundary: |u|^2 = 1e30, never accepted, never uncertain
    const int32_t pad = c.total_j + tid;
    if (pad < ((c.total_j + MF_TILE - 1) & ~(MF_TILE - 1))) {
      L.comp[pad] = 0.f, L.comp[MF_CSTR + pad] = 0.f, L.comp[2 * MF_CSTR + pad] = 0.f;
      L.comp[3 * MF_CSTR + pad] = 1.0e30f;
      L.gid[pad] = 0;
    }
  }
  if (tid < MF_ROWS) L.cnt[tid] = 0;
  mf_stamp(a, tid, 1, t_prev);  // staging (this wave's share: loads returned, LDS written)
  __syncthreads();
  mf_stamp(a, tid, 2, t_prev);  // barrier

  const int32_t ntiles = (c.total_j + MF_TILE - 1) / MF_TILE;
  const int32_t nunits = 2 * ((c.ni + 15) >> 4);
  for (int32_t u = wave; u < nunits; u += MF_WAVES) mf_unit(a, c, L, lane, i_off, (u >> 1) * 16, u & 1, ntiles);
  mf_stamp(a, tid, 3, t_prev);  // search + word stores
  __syncthreads();
  mf_stamp(a, tid, 4, t_prev);  // barrier
  if (tid < c.ni) a.count[a.sorted_row[c.ibeg + tid]] = L.cnt[tid];
  mf_stamp(a, tid, 5, t_prev);  // counts
}

}  // namespace nl
