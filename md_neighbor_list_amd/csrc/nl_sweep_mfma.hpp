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
// Work split: a workgroup (4 waves) owns an i-cell as in k_sweep.  The stencil stream is staged ONCE into LDS as
// four component arrays (ux, uy, uz, |u|^2) + ids.  Every wave handles ALL i-blocks (16 rows each, up to 3 at a
// time) against a contiguous quarter of the j-tiles (16 particles each), so each B operand fetched from LDS feeds
// up to 3 MFMAs.  Hit words (layout MASK_TILE16, 48 per row): word (t / 32)*16 + lam, bit 31 - t % 32 for tile t and
// lam = j % 16.  The waves OR their parts into a row-major LDS image, which leaves as one contiguous block (the
// rows of a cell are consecutive sorted slots); k_fill_masks<T, MASK_TILE16> expands them.
#pragma once

namespace nl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_CAP = SweepCfg<float>::CAP;  // staged particles per cell: the same single-batch limit as the masks
constexpr int MF_CSTR = MF_CAP + 16;          // component array stride: the four bases fall on banks 0, 16, 32, 48
constexpr int MF_TILE = 16;
constexpr int MF_NBMAX = 3;                   // i-blocks (of 16 rows) searched together (mf_rows<1..3>)
constexpr int MF_ROWS = MF_NBMAX * 16;        // rows per pass
constexpr int MF_WORDS = (MF_CAP / MF_TILE + 31) / 32 * 16;  // hit words per row: 3 groups of 32 tiles x 16 lanes = 48

struct MfmaLds {
  float comp[4 * MF_CSTR];  // ux | uy | uz | |u|^2 of the staged stream
  int32_t gid[MF_CAP];
  int32_t cnt[MF_ROWS];
  uint32_t words[MF_ROWS * MF_WORDS];  // hit words of the pass, row-major: leave the kernel as one contiguous block
};
static_assert(sizeof(float) * 4 * MF_CSTR >= sizeof(Pos<float>) * SweepCfg<float>::CAP, "fallback tile fits");
static_assert(MF_WORDS == MASK16_WORDS && MF_WORDS <= WAVE, "k_fill_masks<MASK_TILE16> reads one word per lane");

// first tile and number of tiles of wave w when ntiles are dealt to SWEEP_WAVES waves in contiguous runs
__device__ __forceinline__ void mf_tile_range(int32_t ntiles, int32_t w, int32_t& t_beg, int32_t& nt) {
  const int32_t base = ntiles / SWEEP_WAVES, rem = ntiles % SWEEP_WAVES;
  t_beg = w * base + min(w, rem);
  nt = base + (w < rem ? 1 : 0);
}

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

// NB i-blocks starting at row i0 of the cell against tiles [t_beg, t_beg + nt) of the staged stream.
// i_off: stream position of the cell's own first particle (the i-particles are part of their own stencil, so their
// local coordinates and ids are already in LDS).
template <int NB>
__device__ __forceinline__ void mf_rows(const SweepArgs<float>& a, const CellCtx& c, MfmaLds& L, int lane,
                                        int32_t i_off, int32_t i0, int32_t t_beg, int32_t nt) {
  const int kq = lane >> 4, lam = lane & 15;
  float A[NB];
  f32x4 C[NB];
  int32_t gi[NB][4];
  uint32_t bits[NB][4];
#pragma unroll
  for (int b = 0; b < NB; b++) {
    {
      const int32_t irow = i0 + 16 * b + lam;
      const float u = L.comp[min(kq, 2) * MF_CSTR + i_off + min(irow, c.ni - 1)];
      A[b] = irow < c.ni ? (kq == 3 ? 1.0f : mul_rn(-2.0f, u)) : 0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int32_t irow = i0 + 16 * b + 4 * kq + r;
      const int32_t p = i_off + min(irow, c.ni - 1);
      C[b][r] = irow < c.ni ? sub_rn(L.comp[3 * MF_CSTR + p], a.rc2) : 1.0e30f;  // padding rows: never accepted
      gi[b][r] = L.gid[p];
      bits[b][r] = 0;
    }
  }

  const float* const bp = L.comp + kq * MF_CSTR + lam;
  const int32_t* const gp = L.gid + lam;
  const float delta = a.delta;
  const int32_t t_last = t_beg + nt - 1;

  // One tile's accumulators -> one more bit in every hit word.
  auto process = [&](const f32x4 (&acc)[NB], int32_t gj, int32_t t) {
    // smallest |r2 - rc2| of the tile in this lane: v_min3_f32 with |.| source modifiers, half an instruction per value
    float m = __builtin_fabsf(acc[0][0]);
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = (b == 0 ? 1 : 0); r < 4; r++) m = __builtin_fminf(m, __builtin_fabsf(acc[b][r]));
    // sign(acc) = accepted by distance, sign(gid_i - gid_j) = j is the upper index (ids are >= 0)
#pragma unroll
    for (int b = 0; b < NB; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float av = acc[b][r];  // (bit_cast straight from the vector element picks element 0)
        const uint32_t h = __float_as_uint(av) & (uint32_t)(gi[b][r] - gj);
        bits[b][r] = __builtin_amdgcn_alignbit(bits[b][r], h, 31);  // (bits << 1) | (h >> 31)
      }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(m < delta) != 0, 0)) {
      // rare (about 1 % of the tiles): some |r2 - rc2| is inside the error band of the matrix-core expression.
      // Re-test those elements with the reference's expression on the original coordinates and flip the bit just
      // written where the exact answer differs.  (Nothing here writes the accumulators: the common path keeps them
      // where the MFMA left them, without copies.)
      const int32_t pj_pos = t * MF_TILE + lam;
      const bool jok = pj_pos < c.total_j;
      const Pos<float> pj = a.sorted[jok ? mf_stream_to_sorted(c, pj_pos) : c.ibeg];
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int32_t irow = i0 + 16 * b + 4 * kq + r;
          const float av = acc[b][r];
          if (__builtin_fabsf(av) < delta && jok && irow < c.ni) {
            const Pos<float> pi = a.sorted[c.ibeg + irow];
            const float dx = sub_rn(pj.x, pi.x), dy = sub_rn(pj.y, pi.y), dz = sub_rn(pj.z, pi.z);
            const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
            const bool want = !(r2 > a.rc2), have = (__float_as_uint(av) >> 31) != 0;
            if (want != have && gi[b][r] < gj) bits[b][r] ^= 1u;
            if (a.dbg & 8) atomicAdd(a.dbg_buf + 0, 1ull);
          }
        }
    }
  };

  if (nt > 0) {
    // Two accumulator sets in ping-pong: the MFMAs of the next tile are issued before the vector work of the
    // current one, so the matrix pipe runs underneath it.  Tile indices beyond the wave's share are clamped (the
    // result of such an MFMA is never consumed).
    auto tix = [&](int32_t t) { return min(t, t_last) * MF_TILE; };
    float bv0 = bp[tix(t_beg)], bv1 = bp[tix(t_beg + 1)];
    int32_t g0 = gp[tix(t_beg)], g1 = gp[tix(t_beg + 1)];
    f32x4 acc0[NB], acc1[NB];
#pragma unroll
    for (int b = 0; b < NB; b++) acc0[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b], bv0, C[b], 0, 0, 0);
    for (int32_t t = t_beg; t <= t_last; t += 2) {
#pragma unroll
      for (int b = 0; b < NB; b++) acc1[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b], bv1, C[b], 0, 0, 0);
      bv0 = bp[tix(t + 2)];
      const int32_t g0n = gp[tix(t + 2)];
      process(acc0, g0, t);
#pragma unroll
      for (int b = 0; b < NB; b++) acc0[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[b], bv0, C[b], 0, 0, 0);
      bv1 = bp[tix(t + 3)];
      const int32_t g1n = gp[tix(t + 3)];
      if (t + 1 <= t_last) process(acc1, g1, t + 1);
      g0 = g0n, g1 = g1n;
    }
  }

  // Deposit the hit words in the row-major LDS image (tile t -> group t / 32, bit 31 - t % 32: a wave's share
  // spans at most two groups) and add the popcounts to the row counts.
  const uint32_t sh_hi = 64u - (uint32_t)nt, sh_lo = (uint32_t)t_beg & 31u;
  const int32_t grp = t_beg >> 5;
#pragma unroll
  for (int b = 0; b < NB; b++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int32_t lrow = 16 * b + 4 * kq + r;
      const uint64_t v64 = nt > 0 ? ((uint64_t)bits[b][r] << sh_hi) >> sh_lo : 0ull;
      const uint32_t hi = (uint32_t)(v64 >> 32), lo = (uint32_t)v64;
      if (hi) atomicOr(&L.words[lrow * MF_WORDS + grp * 16 + lam], hi);
      if (lo) atomicOr(&L.words[lrow * MF_WORDS + (grp + 1) * 16 + lam], lo);
      int32_t v = __popc(hi) + __popc(lo);
      v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1 within the 16 lanes of a row group
      v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
      v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
      if (lam == 15 && v) atomicAdd(&L.cnt[lrow], v);
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

__global__ void __launch_bounds__(SWEEP_WAVES* WAVE, 4) k_sweep_mfma_f32(SweepArgs<float> a) {
  __shared__ __attribute__((aligned(16))) MfmaLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long t_prev = 0;
  if (a.dbg & 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  CellCtx c;
  if (!cell_setup(a, lane, c)) return;
  mf_stamp(a, tid, 0, t_prev);  // cell table
  if (c.total_j > MF_CAP) {
    // stencil larger than one LDS batch: counts only, by the VALU search (k_fill_masks searches such cells again)
    cell_search<float, MODE_COUNT>(a, c, reinterpret_cast<Pos<float>*>(L.comp), tid, lane, wave);
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

  // ---- stage the stream as component arrays
  for (int32_t sg = wave; sg < NSEG; sg += SWEEP_WAVES) {
    const int32_t len = __builtin_amdgcn_readlane(c.seg_len, sg);
    if (len == 0) continue;
    const int32_t src = __builtin_amdgcn_readlane(c.seg_src, sg);
    const int32_t off = __builtin_amdgcn_readlane(c.seg_off, sg);
    if (a.dbg & 2) continue;  // diagnostics: no staging copies
    for (int32_t k = lane; k < len; k += 2 * WAVE) {
      const int32_t k1 = k + WAVE;
      const bool p1 = k1 < len;
      const Pos<float> v0 = a.sorted[src + k];
      const Pos<float> v1 = a.sorted[src + (p1 ? k1 : k)];
      {
        const float ux = sub_rn(v0.x, ccx), uy = sub_rn(v0.y, ccy), uz = sub_rn(v0.z, ccz);
        L.comp[off + k] = ux, L.comp[MF_CSTR + off + k] = uy, L.comp[2 * MF_CSTR + off + k] = uz;
        L.comp[3 * MF_CSTR + off + k] = add_rn(add_rn(mul_rn(ux, ux), mul_rn(uy, uy)), mul_rn(uz, uz));
        L.gid[off + k] = v0.gid;
      }
      if (p1) {
        const float ux = sub_rn(v1.x, ccx), uy = sub_rn(v1.y, ccy), uz = sub_rn(v1.z, ccz);
        L.comp[off + k1] = ux, L.comp[MF_CSTR + off + k1] = uy, L.comp[2 * MF_CSTR + off + k1] = uz;
        L.comp[3 * MF_CSTR + off + k1] = add_rn(add_rn(mul_rn(ux, ux), mul_rn(uy, uy)), mul_rn(uz, uz));
        L.gid[off + k1] = v1.gid;
      }
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
  uint4* const words4 = reinterpret_cast<uint4*>(L.words);
  constexpr int NW4 = MF_ROWS * MF_WORDS / 4;
  for (int k = tid; k < NW4; k += SWEEP_WAVES * WAVE) words4[k] = make_uint4(0, 0, 0, 0);
  if (tid < MF_ROWS) L.cnt[tid] = 0;
  mf_stamp(a, tid, 1, t_prev);  // staging (this wave's share: loads returned, LDS written)
  __syncthreads();
  mf_stamp(a, tid, 2, t_prev);  // barrier

  const int32_t ntiles = (c.total_j + MF_TILE - 1) / MF_TILE;
  int32_t t_beg, nt;
  mf_tile_range(ntiles, wave, t_beg, nt);
  if (a.dbg & 1) nt = 0;  // diagnostics: staging, row setup and mask stores only
  for (int32_t i0 = 0; i0 < c.ni; i0 += MF_ROWS) {
    const int32_t rows = min(MF_ROWS, c.ni - i0);
    switch ((rows + 15) >> 4) {
      case 1: mf_rows<1>(a, c, L, lane, i_off, i0, t_beg, nt); break;
      case 2: mf_rows<2>(a, c, L, lane, i_off, i0, t_beg, nt); break;
      default: mf_rows<3>(a, c, L, lane, i_off, i0, t_beg, nt); break;
    }
    mf_stamp(a, tid, 3, t_prev);  // search + deposit
    __syncthreads();
    mf_stamp(a, tid, 4, t_prev);  // barrier
    // the pass's rows are consecutive sorted slots: their hit words leave as one contiguous block
    uint4* const out4 = reinterpret_cast<uint4*>(a.masks + (size_t)(c.ibeg + i0) * MF_WORDS);
    const bool more = i0 + MF_ROWS < c.ni;
    for (int k = tid; k < rows * (MF_WORDS / 4); k += SWEEP_WAVES * WAVE) {
      out4[k] = words4[k];
      if (more) words4[k] = make_uint4(0, 0, 0, 0);
    }
    if (tid < rows) {
      a.count[a.sorted_row[c.ibeg + i0 + tid]] = L.cnt[tid];
      L.cnt[tid] = 0;
    }
    mf_stamp(a, tid, 5, t_prev);  // stores (until acknowledged: only the diagnostic waits for them)
    if (more) __syncthreads();
  }
}

}  // namespace nl
