// nl_lean.hpp -- the fp32 open-box COUNT_MASKS sweep for cells whose stream fits one LDS buffer (k_sweep_lean_f32: a
// workgroup per cell, the stream by LDS-DMA, the groups' i-particles taken from the staged stream) and the batched
// search of the cells it hands over (k_sweep_list_f32).  The coarse-cell path: builds that cannot take the fine-row
// search of nl_rows.hpp (full-stencil boxes whose cell edge equals the cut-off, lists of the few overflowing cells).
// (Round 2 also had two persistent forms of this sweep -- a software pipeline over cells -- measured 13-20 % slower:
// tools/experiments/half_shell_and_persistent_sweeps.patch, profiles/r02_count_sweep_investigation.txt.)
#pragma once

namespace nl {

constexpr int PIPE_G = 7;  // most i-particles of a group: one group per wave up to 28 particles per cell

// The stream of cell c into `dst`, asynchronously: wave v takes segments v, v + NW, v + 2 NW ...  Lane l's 16 bytes land at
// the (wave-uniform) LDS address + 16 l.  The sentinels up to the tile boundary are ordinary LDS writes.
template <int NW = SWEEP_WAVES>
__device__ __forceinline__ void pipe_stage(const SweepArgs<float>& a, const CellCtx& c, Pos<float>* dst, int tid, int lane, int wave) {
#pragma unroll 1
  for (int sg = wave; sg < NSEG; sg += NW) {
    const int32_t len = __builtin_amdgcn_readlane(c.seg_len, sg);
    const int32_t src = __builtin_amdgcn_readlane(c.seg_src, sg);
    const int32_t off = __builtin_amdgcn_readlane(c.seg_off, sg);
#pragma unroll 1
    for (int32_t kb = 0; kb < len; kb += WAVE) {
      if (kb + lane < len)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.sorted + src + kb + lane),
                                         (__attribute__((address_space(3))) void*)(dst + off + kb), 16, 0, 0);
    }
  }
  const int32_t pad = c.total_j + tid;
  if (pad < ((c.total_j + WAVE - 1) & ~(WAVE - 1))) {
    Pos<float> sentinel;  // far outside any box (finite r2, never in range) and an id that is never the upper one
    sentinel.x = 1.0e18f, sentinel.y = 0.f, sentinel.z = 0.f, sentinel.gid = INT32_MIN;
    dst[pad] = sentinel;
  }
}

// What a wave still has to store for the last group it searched.
struct PipePending {
  uint32_t w[PIPE_G];
  int32_t mine, row_l, slot0, gcount;  // gcount == 0: nothing pending
};

template <bool NT>
__device__ __forceinline__ void pipe_flush(const SweepArgs<float>& a, PipePending& p, int lane, bool hi) {
  if (p.gcount <= 0) return;  // uniform
  // (the count first: its row index is the one load this waits for, and a wait placed behind the mask stores would be
  // a wait for them as well)
  if (lane < p.gcount) a.count[p.row_l] = p.mine;
#pragma unroll
  for (int k = 0; k < PIPE_G; k++)
    if (k < p.gcount) mask_store<NT>(a.masks, a.masks_hi, (size_t)(p.slot0 + k) * a.mask_nb, lane, p.w[k], hi);
  p.gcount = 0;
}

template <bool FULL, int GC>
__device__ __forceinline__ void pipe_group(const SweepArgs<float>& a, const Pos<float>* tile, int32_t nj, int32_t ntiles, int lane,
                                           const Pos<float>& pi_l, int32_t slot0, int32_t self0, PipePending& p) {
  uint32_t words[GC];
  p.mine = search_group<float, MODE_COUNT_MASKS, GC, FULL, FULL, false, false>(a, tile, nj, ntiles, lane, pi_l, 0, slot0, true, self0,
                                                                             0, nullptr, 0.f, 0.f, 0.f, words);
#pragma unroll
  for (int k = 0; k < GC; k++) p.w[k] = words[k];
}

// The pair search of the single-batch cell c, whose stream is in `tile`.
template <bool FULL, int NW = SWEEP_WAVES>
__device__ __forceinline__ void pipe_search(const SweepArgs<float>& a, const CellCtx& c, const Pos<float>* tile, int lane, int wave,
                                            PipePending& p) {
  constexpr int G = PIPE_G;
  const int32_t ibeg = c.ibeg, ni = c.ni, nj = c.total_j;
  const int32_t ntiles = (nj + WAVE - 1) / WAVE;
  // groups: `rounds` per wave, the i-particles spread evenly over them (sizes differ by at most one)
  const int32_t rounds = (ni + NW * G - 1) / (NW * G);
  const int32_t ngroups = rounds * NW;
  const int32_t gbase = ni / ngroups, grem = ni - gbase * ngroups;
  // where the cell's own particles sit in the stream: in the (dz,dy) = (0,0) row, first or wrapped x-part
  const int32_t s4 = __builtin_amdgcn_readlane(c.seg_src, 4), l4 = __builtin_amdgcn_readlane(c.seg_len, 4);
  const bool in4 = ibeg >= s4 && ibeg < s4 + l4;
  const int32_t own = in4 ? __builtin_amdgcn_readlane(c.seg_off, 4) + ibeg - s4
                          : __builtin_amdgcn_readlane(c.seg_off, 13) + ibeg - __builtin_amdgcn_readlane(c.seg_src, 13);
  for (int32_t g = wave; g < ngroups; g += NW) {
    const int32_t i0 = g * gbase + min(g, grem);
    const int32_t gcount = gbase + (g < grem ? 1 : 0);  // wave-uniform
    if (gcount <= 0) break;
    pipe_flush<NW == SWEEP_WAVES>(a, p, lane, hi_plane_used(a.mask_nb, ntiles));  // (a wave with several groups: the previous one's words go out before the next search)
    const int32_t k = min(lane, gcount - 1);
    Pos<float> pi_l = tile[own + i0 + k];  // the group's i-particles come from the staged stream, not from memory
    const int32_t row_l = a.sorted_row[ibeg + i0 + k];
    if (lane >= gcount) pi_l.x = 0, pi_l.y = 0, pi_l.z = 0, pi_l.gid = 0;
    const int32_t slot0 = ibeg + i0, self0 = own + i0;
    switch (gcount) {
      case 1: pipe_group<FULL, 1>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      case 2: pipe_group<FULL, 2>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      case 3: pipe_group<FULL, 3>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      case 4: pipe_group<FULL, 4>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      case 5: pipe_group<FULL, 5>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      case 6: pipe_group<FULL, 6>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
      default: pipe_group<FULL, 7>(a, tile, nj, ntiles, lane, pi_l, slot0, self0, p); break;
    }
    p.row_l = row_l, p.slot0 = slot0, p.gcount = gcount;
  }
}

// One workgroup per cell, like k_sweep_count_masks_f32, but only what a single-batch cell needs: the stream by LDS-DMA,
// the groups' i-particles taken from the staged stream (no loads of their own), no batch loop, no progress words; a
// cell whose stream does not fit goes on the list of k_sweep_list_f32.  (NL_PIPE=1.)
// NW, CAP: 4 waves and the full buffer (20 KB: 8 workgroups = 32 waves per CU), or -- boxes whose streams stay below
// half of it: BASELINE config 3 -- 2 waves and half the buffer (10 KB: 16 workgroups = 32 waves per CU): the same
// number of group passes per cell by half as many waves, i.e. half as many cell tables, barriers and flushes (what a
// wave does besides searching is half of its life when its search is one pass of 9 tiles).
constexpr int LEAN_SMALL_CAP = SweepCfg<float>::CAP / 2;
template <bool FULL, int NW = SWEEP_WAVES, int CAP = SweepCfg<float>::CAP>
__global__ void __launch_bounds__(NW* WAVE, 8) __attribute__((amdgpu_num_sgpr(80))) k_sweep_lean_f32(SweepArgs<float> a) {
  __shared__ __attribute__((aligned(32))) Pos<float> buf[CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  __builtin_amdgcn_s_setprio(NL_PRIO);  // everything but the tile loop of search_group
  CellCtx c;
  if (!cell_setup(a, lane, c)) return;
  if (c.total_j > CAP) {  // (rare) several LDS batches: k_sweep_list_f32
    if (tid == 0) a.full27_list[atomicAdd(a.full27_count, 1)] = c.cx + (c.cy + c.cz * a.my) * a.mx;
    return;
  }
  pipe_stage<NW>(a, c, buf, tid, lane, wave);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  PipePending pend;
  pend.gcount = 0, pend.mine = 0, pend.row_l = 0, pend.slot0 = 0;
  pipe_search<FULL, NW>(a, c, buf, lane, wave, pend);
  pipe_flush<NW == SWEEP_WAVES>(a, pend, lane, hi_plane_used(a.mask_nb, (c.total_j + WAVE - 1) / WAVE));
}

// The cells k_sweep_pipe_f32 left out (local cell indices in full27_list): the batched search, a workgroup per cell.
template <bool FULL>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE) __attribute__((amdgpu_num_sgpr(80))) k_sweep_list_f32(SweepArgs<float> a) {
  constexpr int CAP = SweepCfg<float>::CAP;
  __shared__ __attribute__((aligned(32))) Pos<float> tile[CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int32_t count = *a.full27_count;
  for (int32_t idx = blockIdx.x; idx < count; idx += gridDim.x) {
    if (idx != (int32_t)blockIdx.x) __syncthreads();  // everyone is done with the previous cell's LDS
    const int32_t cell = a.full27_list[idx];
    const int32_t row = cell / a.mx, cx = cell - row * a.mx, cz = row / a.my, cy = row - cz * a.my;
    CellCtx c;
    if (!cell_setup_at(a, lane, cx, cy, cz, c)) continue;
    cell_search<float, MODE_COUNT_MASKS, CAP, SWEEP_WAVES, FULL, false, false>(a, c, tile, tid, lane, wave);
  }
}

}  // namespace nl
