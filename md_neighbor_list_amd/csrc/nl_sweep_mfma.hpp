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
// Work split: a workgroup of 4 waves owns an i-cell.  The stencil stream is staged ONCE into LDS as four component
// arrays (ux, uy, uz, |u|^2) + ids (26 KB: six workgroups = 24 waves per CU).  A unit of work is one i-block (16
// rows) against the tiles (16 staged particles each) of one residue class t = w (mod 4); wave w walks the i-blocks
// with residue w, so the four waves -- one per SIMD -- do the same number of steps.  Little per-wave state (one MFMA
// accumulator pair, 4 row constants, 4 hit words).
// Hit words: step s of residue w tests staged particle s*64 + (w*16 + lam), which is bit s of word w*16 + lam of the
// mask layout search_group writes: k_fill_masks expands both alike, and the units own 16 words each (64-byte stores).
#pragma once

namespace nl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MF_CAP = 1152;                  // staged particles per cell on this path (18 x 64); longer streams (one
                                              // cell in 1500 at rho = 1) take the VALU search
constexpr int MF_PAD = 144;                   // the operand fetch runs up to two steps (8 tiles) past the last tile
constexpr int MF_CSTR = MF_CAP + MF_PAD;      // component array stride: the four bases fall on banks 0, 16, 32, 48
constexpr int MF_TILE = 16;
constexpr int MF_WAVES = 4;                   // waves per workgroup: wave w takes the tiles t = w (mod 4)
constexpr int MF_ROWS = 64;                   // most rows of a cell on this path (LDS row counters); fuller cells take
                                              // the VALU search, which writes the same masks

struct MfmaLds {
  float comp[4 * MF_CSTR];  // ux | uy | uz | |u|^2 of the staged stream
  int32_t gid[MF_CSTR];
  int32_t cnt[MF_ROWS];
};
static_assert(sizeof(float) * 4 * MF_CSTR >= sizeof(Pos<float>) * SweepCfg<float>::CAP, "fallback tile fits");
static_assert(MF_CAP / MF_TILE / 4 <= 32 && MF_CSTR % 64 == 16 && MF_CAP <= SweepCfg<float>::CAP, "layout");

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

// A 16-byte load that the compiler does not see as a pending memory operation (the wait is inside).  Used on the
// rare exact re-test only: with ordinary loads there the compiler guards the step loop's head with s_waitcnt
// vmcnt(0), which makes every unit wait for the hit-word STORES of the previous one.
__device__ __forceinline__ Pos<float> mf_load_pos_blocking(const Pos<float>* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  Pos<float> r;
  r.x = v[0], r.y = v[1], r.z = v[2], r.gid = (int32_t)__float_as_uint(v[3]);
  return r;
}

#ifdef MF_ABL_NO_MFMA  // ablation: the vector work on an accumulator that no MFMA produces (bv keeps the loads alive)
#define MF_MFMA(A_, B_, C_) ((C_) + (B_))
#else
#define MF_MFMA(A_, B_, C_) __builtin_amdgcn_mfma_f32_16x16x4f32(A_, B_, C_, 0, 0, 0)
#endif

// ---- the f16 form (NL_SWEEP_VARIANT=5).  The fp32 MFMA runs at the vector rate and holds the SIMD's vector issue
// for its 32 cycles; v_mfma_f32_16x16x32_f16 takes 8.  An fp32 value v (scaled by a power of two so that |v| < 2^7
// for everything near the cell) is carried as two halves, h = v with its low 13 mantissa bits cleared and
// l = rtz_f16(v - h): products of halves are exact in the fp32 accumulator, and h + l differs from v by < 2^-21 |v|.
// With K = 32 one instruction holds, per coordinate, the four products (-2 h_i)(h_j), (-2 h_i)(l_j), (-2 l_i)(h_j),
// (-2 l_i)(l_j), and 1 * (|u_j|^2 as two halves); lanes 16 q .. 16 q + 15 carry k = 8 q .. 8 q + 7 (q = x, y, z, norm).
// The B operand of a lane is {(h_j, l_j), (h_j, l_j), 0, 0}: one packed dword per staged particle and component, as
// in the fp32 form.  Decisive values only, the same band logic (delta is widened by the host for the coarser split).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mf_split16(float v) {  // (h, l) packed: h in the low half
  const float h = __uint_as_float(__float_as_uint(v) & 0xFFFFE000u);
  typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
  const fp16x2 p = __builtin_amdgcn_cvt_pkrtz(h, v - h);
  return __builtin_bit_cast(uint32_t, p);
}
__device__ __forceinline__ float mf_join16(uint32_t p) {  // h + l back as fp32
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  const f16x2 v = __builtin_bit_cast(f16x2, p);
  return (float)v[0] + (float)v[1];
}

// One unit: the i-block of 16 rows starting at row i0 of the cell against the tiles t = res (mod 4) of the ntiles
// staged tiles.
// i_off: stream position of the cell's own first particle (the i-particles are part of their own stencil, so their
// local coordinates and ids are already in LDS).
template <bool F16 = false, bool FULL = false>
__device__ __forceinline__ void mf_unit(const SweepArgs<float>& a, const CellCtx& c, MfmaLds& L, int lane,
                                        int32_t i_off, int32_t i0, int32_t res, int32_t ntiles) {
  const int kq = lane >> 4, lam = lane & 15;
  float A = 0.f;      // fp32 form: one value of the 16 x 4 operand
  f16x8 A16 = {};     // f16 form: eight halves of the 16 x 32 operand
  f32x4 C;
  int32_t gi[4];
  {
    const int32_t irow = i0 + lam;
    const float u = L.comp[min(kq, 2) * MF_CSTR + i_off + min(irow, c.ni - 1)];
    if constexpr (!F16) {
      A = irow < c.ni ? (kq == 3 ? 1.0f : mul_rn(-2.0f, u)) : 0.0f;
    } else {
      typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
      const f16x2 hl = __builtin_bit_cast(f16x2, __float_as_uint(u));  // (h, l) of this coordinate
      const _Float16 m2 = (_Float16)-2.0f, one = (_Float16)1.0f, zero = (_Float16)0.0f;
      const _Float16 a_h = kq == 3 ? one : m2 * hl[0], a_l = kq == 3 ? zero : m2 * hl[1];
      if (irow < c.ni) A16 = f16x8{a_h, a_h, a_l, a_l, zero, zero, zero, zero};
    }
  }
  const float rc2s = F16 ? a.rc2 * a.mf_scale2 : a.rc2;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int32_t irow = i0 + 4 * kq + r;
    const int32_t p = i_off + min(irow, c.ni - 1);
    const float n2 = F16 ? mf_join16(__float_as_uint(L.comp[3 * MF_CSTR + p])) : L.comp[3 * MF_CSTR + p];
    C[r] = irow < c.ni ? sub_rn(n2, rc2s) : 1.0e30f;  // padding rows: never accepted
    gi[r] = L.gid[p];
  }

  const float* const bp = L.comp + kq * MF_CSTR + lam;
  const int32_t* const gp = L.gid + lam;
  const float delta = F16 ? a.delta16 : a.delta;
  uint32_t bits[4];
  // acc = MFMA(A, staged dword, C)
  auto mma = [&](float bv) -> f32x4 {
    if constexpr (!F16) {
      return MF_MFMA(A, bv, C);
    } else {
      // Written as assembly for its VGPR form: through the builtin this compiler allocates the accumulator in AGPRs
      // (4 v_accvgpr_read per step and one accumulator for both tiles of the ping-pong).  The compiler does not
      // see the MFMA -> VALU read hazard of an asm statement; the reads of the result come a whole process() of the
      // OTHER tile later (>= 16 vector instructions behind a branch, i.e. a block boundary), far more than the 4
      // passes the instruction takes.  The other hazard is on the way in: the B tuple is assembled by v_mov right
      // before, and an MFMA must not read a VGPR a VALU instruction has just written -- hence the s_nop (without
      // it the kernel produced wrong lists; 5 wait states verified).
      const uint32_t w = __float_as_uint(bv);
      const u32x4 b4 = {w, w, 0u, 0u};
      f32x4 d;
      asm volatile("s_nop 4\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "v"(A16), "v"(b4), "v"(C));
      return d;
    }
  };

  // One tile's accumulators -> one more bit in every hit word; returns the smallest |r2 - rc2| this lane saw
  // (v_min3_f32 with |.| source modifiers: half an instruction per value).
  auto process = [&](const f32x4& acc, int32_t gj, float m) {
#pragma unroll
    for (int r = 0; r < 4; r++) m = __builtin_fminf(m, __builtin_fabsf(acc[r]));
    // sign(acc) = accepted by distance, sign(gid_i - gid_j) = j is the upper index (ids are >= 0)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float av = acc[r];  // (bit_cast straight from the vector element picks element 0)
      // (full list: everyone in range but the row itself, whose bit is cleared after the loop: no id work per test)
      const uint32_t h = FULL ? __float_as_uint(av) : __float_as_uint(av) & (uint32_t)(gi[r] - gj);
      bits[r] = __builtin_amdgcn_alignbit(bits[r], h, 31);  // (bits << 1) | (h >> 31)
    }
    return m;
  };
  // Rare (about 2 % of the step pairs): some |r2 - rc2| of tile t was inside the error band of the matrix-core
  // expression.  The tile's 16 x 16 block is decided again from the original coordinates with the reference's
  // expression and its bit (position `pos` of the words by now) rewritten.  Nothing here touches the accumulators.
  auto retest = [&](int32_t t, int32_t gj, uint32_t pos) {
    const int32_t pj_pos = t * MF_TILE + lam;
    const bool jok = pj_pos < c.total_j;
    const Pos<float> pj = mf_load_pos_blocking(a.sorted + (jok ? mf_stream_to_sorted(c, pj_pos) : c.ibeg));
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int32_t irow = i0 + 4 * kq + r;
      const Pos<float> pi = mf_load_pos_blocking(a.sorted + c.ibeg + min(irow, c.ni - 1));
      const float dx = sub_rn(pj.x, pi.x), dy = sub_rn(pj.y, pi.y), dz = sub_rn(pj.z, pi.z);
      const float r2 = add_rn(add_rn(mul_rn(dx, dx), mul_rn(dy, dy)), mul_rn(dz, dz));
      const bool want = !(r2 > a.rc2) && (FULL ? gi[r] != pj.gid : gi[r] < gj) && jok && irow < c.ni;
      bits[r] = (bits[r] & ~(1u << pos)) | ((want ? 1u : 0u) << pos);
    }
    if (a.dbg & 8) atomicAdd(a.dbg_buf + 0, 1ull);
  };

  // The unit's tiles are t = res, res + 4, res + 8, ...: step s tests the 16 staged particles (res + 4 s)*16 + lam,
  // i.e. particle s*64 + (res*16 + lam) -- word res*16 + lam, bit s of the mask layout that k_fill_masks expands
  // (search_group writes the same words): the four units of an i-block own 16 words each.
  const int32_t ns = a.dbg & 1 ? 0 : (ntiles - res + 3) >> 2;  // steps of this unit (<= 18)
#pragma unroll
  for (int r = 0; r < 4; r++) bits[r] = 0;
  if (ns > 0) {
    // Two steps per trip, two accumulators: the MFMA of step s + 2 is issued as soon as step s has been consumed, so
    // it runs under the vector work of step s + 1.  Operands are fetched two steps ahead with one running pointer
    // (consecutive steps of a wave are 64 dwords apart: ds_read2st64_b32); the fetches past the last step read the
    // pad of the arrays, and the MFMAs fed with them are never consumed.
    constexpr int STRIDE = 4 * MF_TILE;
    const float* pb = bp + res * MF_TILE;
    const int32_t* pg = gp + res * MF_TILE;
    float b0 = pb[0], b1 = pb[STRIDE];
    int32_t g0 = FULL ? 0 : pg[0], g1 = FULL ? 0 : pg[STRIDE];
    f32x4 acc0 = mma(b0), acc1 = mma(b1);
    for (int32_t s_ = 0; s_ < ns; s_ += 2) {
      pb += 2 * STRIDE, pg += 2 * STRIDE;
#ifdef MF_ABL_NO_LDS  // ablation: every step reads the same tile (the loads are hoisted out of the loop)
      const float nb0 = b0, nb1 = b1;
      const int32_t ng0 = g0, ng1 = g1;
#else
      const float nb0 = pb[0], nb1 = pb[STRIDE];
      const int32_t ng0 = FULL ? 0 : pg[0], ng1 = FULL ? 0 : pg[STRIDE];
#endif
      const bool two = s_ + 1 < ns;
      float m = process(acc0, g0, __builtin_huge_valf());
      acc0 = mma(nb0);
      if (two) m = process(acc1, g1, m);
      acc1 = mma(nb1);
#ifndef MF_ABL_NO_UNC  // tools/mfma_bench ablation: no band check at all (timing only, results wrong)
      if (__builtin_expect(__builtin_amdgcn_ballot_w64(m < delta) != 0, 0)) {
        retest(res + 4 * s_, g0, two ? 1u : 0u);
        if (two) retest(res + 4 * (s_ + 1), g1, 0u);
      }
#endif
      g0 = ng0, g1 = ng1;
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int32_t irow = i0 + 4 * kq + r;
    // step s sits at bit ns - 1 - s (v_alignbit shifts left): reverse into bit s
    uint32_t w = ns > 0 ? __brev(bits[r]) >> (32 - ns) : 0u;
    if (FULL) {  // the row's own particle (distance 0: always "in range") is staged at i_off + irow: tile, lane, step
      const int32_t p_self = i_off + irow, t_self = p_self >> 4;
      if ((t_self & 3) == res && (p_self & 15) == lam) w &= ~(1u << (t_self >> 2));
    }
#ifndef MF_ABL_NO_STORE
    if (irow < c.ni) mask_store(a.masks, (size_t)(c.ibeg + irow), res * 16 + lam, w);
#endif
    int32_t v = __popc(w);
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

template <bool F16, bool FULL = false>
__device__ __forceinline__ void mf_cell(const SweepArgs<float>& a) {
  __shared__ __attribute__((aligned(16))) MfmaLds L;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned long long t_prev = 0;
  if (a.dbg & 4) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  CellCtx c;
  if (!cell_setup(a, lane, c)) return;
  mf_stamp(a, tid, 0, t_prev);  // cell table
  if (c.total_j > MF_CAP || c.ni > MF_ROWS) {
    // stencil larger than one LDS batch, or a very full cell: the VALU search (same masks, same counts; cells of
    // several batches get counts only and are searched again by k_fill_masks)
    cell_search<float, MODE_COUNT_MASKS, SweepCfg<float>::CAP, MF_WAVES, FULL>(a, c, reinterpret_cast<Pos<float>*>(L.comp),
                                                                               tid, lane, wave);
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
  // the far sentinel of the f16 form: |u|^2 = 60000 (as halves), coordinates 0: acc = c_i + 60000 > 0 for every row
  const uint32_t far16 = mf_split16(60000.0f);
  auto put = [&](int32_t p, const Pos<float>& v) {
    float ux = sub_rn(v.x, ccx), uy = sub_rn(v.y, ccy), uz = sub_rn(v.z, ccz);
    if constexpr (!F16) {
      L.comp[p] = ux, L.comp[MF_CSTR + p] = uy, L.comp[2 * MF_CSTR + p] = uz;
      L.comp[3 * MF_CSTR + p] = add_rn(add_rn(mul_rn(ux, ux), mul_rn(uy, uy)), mul_rn(uz, uz));
    } else {
      ux *= a.mf_scale, uy *= a.mf_scale, uz *= a.mf_scale;  // power of two: exact
      const float n2 = add_rn(add_rn(mul_rn(ux, ux), mul_rn(uy, uy)), mul_rn(uz, uz));
      // a particle this far from the cell centre (|u| scale >= 245, against <= 42 for the cell's own particles and
      // rc scale <= 48) cannot be within the cut-off of any row near the centre; its halves would overflow: it
      // becomes a far sentinel.  (Rows that are themselves far from the centre: see the check after the barrier.)
      const bool far = !(n2 < 60000.0f);
      L.comp[p] = __uint_as_float(far ? 0u : mf_split16(ux));
      L.comp[MF_CSTR + p] = __uint_as_float(far ? 0u : mf_split16(uy));
      L.comp[2 * MF_CSTR + p] = __uint_as_float(far ? 0u : mf_split16(uz));
      L.comp[3 * MF_CSTR + p] = __uint_as_float(far ? far16 : mf_split16(n2));
    }
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
      L.comp[3 * MF_CSTR + pad] = F16 ? __uint_as_float(far16) : 1.0e30f;
      L.gid[pad] = 0;
    }
  }
  if (tid < MF_ROWS) L.cnt[tid] = 0;
  mf_stamp(a, tid, 1, t_prev);  // staging (this wave's share: loads returned, LDS written)
  __syncthreads();
  mf_stamp(a, tid, 2, t_prev);  // barrier

  const int32_t ntiles = (c.total_j + MF_TILE - 1) / MF_TILE;
  if constexpr (F16) {
    // A particle can lie far from the cell it is filed in: a coordinate that rounds up to the box edge is wrapped to
    // cell 0 (GenHash + ApplyPBC, neighlist_cpu.hpp:51-66), coordinates up to one box length outside are legal.
    // As a j-particle such a one is a far sentinel, which is right for every row near the cell centre; as a ROW it
    // needs the real arithmetic: a cell that owns one is searched by the VALU path (same masks, same counts).
    // Every wave tests all rows itself, so the decision is uniform without another barrier.
    const bool row_far = lane < c.ni && mf_join16(__float_as_uint(L.comp[3 * MF_CSTR + i_off + lane])) >= 30000.0f;
    if (__builtin_amdgcn_ballot_w64(row_far) != 0) {
      __syncthreads();  // (rare) everyone has read its rows before the stream is staged again as positions
      cell_search<float, MODE_COUNT_MASKS, SweepCfg<float>::CAP, MF_WAVES, FULL>(a, c, reinterpret_cast<Pos<float>*>(L.comp),
                                                                                 tid, lane, wave);
      return;
    }
  }
  // one unit per (i-block, tile residue): wave w walks the i-blocks with residue w -- every wave of the workgroup
  // (one per SIMD) does the same number of steps
  for (int32_t i0 = 0; i0 < c.ni; i0 += 16) mf_unit<F16, FULL>(a, c, L, lane, i_off, i0, wave, ntiles);
  mf_stamp(a, tid, 3, t_prev);  // search + word stores
  __syncthreads();
  mf_stamp(a, tid, 4, t_prev);  // barrier
  if (tid < c.ni) a.count[a.sorted_row[c.ibeg + tid]] = L.cnt[tid];
  mf_stamp(a, tid, 5, t_prev);  // counts
}

__global__ void __launch_bounds__(MF_WAVES* WAVE, 6) k_sweep_mfma_f32(SweepArgs<float> a) { mf_cell<false>(a); }
__global__ void __launch_bounds__(MF_WAVES* WAVE, 6) k_sweep_mfma_f16(SweepArgs<float> a) { mf_cell<true>(a); }
// the full list (both directions): no id test per pair, the row's own bit is cleared at the end
__global__ void __launch_bounds__(MF_WAVES* WAVE, 6) k_sweep_mfma_f16_full(SweepArgs<float> a) { mf_cell<true, true>(a); }

}  // namespace nl
