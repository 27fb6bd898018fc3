// nl_pipe.hpp -- the fp32 open-box COUNT_MASKS sweep for cells whose stream fits one LDS buffer: k_sweep_lean_f32 (the
// default: a workgroup per cell) at the end of this file, and first its two persistent forms, a software pipeline over cells.
//
// k_sweep_count_masks_f32 gives every cell its own workgroup, and a wave of it spends more than half of its life
// outside the pair search: kernel arguments, the cell's segment table (two dependent loads), the stencil stream
// (two or three dependent round trips per wave), a barrier -- tools/count_phases.py: 27 000 of its 50 000 cycles at
// BASELINE config 2.  Eight waves per SIMD cannot hide that: on average three of them are searching.
//
// Here a workgroup of 8 waves is PERSISTENT: it walks a run of consecutive cells, and while it searches cell w in one
// LDS buffer
//   - the stencil stream of cell w + 1 is on its way into the other buffer by LDS-DMA (global_load_lds_dwordx4: no
//     registers, nothing to wait for until the next barrier), and
//   - the segment table of cell w + 2 is being loaded into registers.
// One barrier per cell (the stream of the next cell has landed / everyone has left the buffer the cell after it will
// use).  The hit words and counts of a wave's last group are stored after that barrier, so that the wait in front of
// it (vmcnt(0): the DMA) never waits for a store that was issued a moment ago.
// A cell whose stream does not fit one buffer (never at the densities this path is selected for, short of a cluster)
// is put on a list and searched afterwards by k_sweep_list_f32, the batched search of the one-cell-per-workgroup
// kernels: inlined here, its registers would be this kernel's.
//
// Same tests, same words, same counts as k_sweep_count_masks_f32: the search itself is search_group.
#pragma once

namespace nl {

constexpr int PIPE_WAVES = 8;
constexpr int PIPE_CAP = 1216;  // staged particles per buffer: two buffers of 19 KB, four workgroups (32 waves) per CU
constexpr int PIPE_G = 7;  // most i-particles of a group: one group per wave up to 56 particles per cell

// A cell's table loads, issued and not yet waited for.
struct PipeRaw {
  int32_t ibeg, iend;        // cell_start[cell], cell_start[cell + 1] (every lane loads the same two words)
  int32_t seg_src, seg_end;  // lane s < NSEG: cell_start at the two ends of segment s
  int32_t cx, cy, cz;
};

// w: index of the cell among the cells of the launch (x fastest); !valid: past the end of this workgroup's run.
__device__ __forceinline__ void pipe_issue(const SweepArgs<float>& a, int lane, int32_t w, bool valid, PipeRaw& r) {
  const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), cx = w - wy * a.mx;
  const int32_t wz = (int32_t)fastdiv((uint32_t)wy, a.div_my), cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
  r.cx = cx, r.cy = cy, r.cz = cz;
  r.ibeg = 0, r.iend = 0, r.seg_src = 0, r.seg_end = 0;
  if (!valid) return;  // uniform
  const int32_t cell = cx + (cy + cz * a.my) * a.mx;
  r.ibeg = a.cell_start[cell];
  r.iend = a.cell_start[cell + 1];
  if (lane < NSEG) {
    int32_t i0, i1, wrap;
    segment_cells(a, lane, cx, cy, cz, i0, i1, wrap);
    r.seg_src = a.cell_start[i0];
    r.seg_end = a.cell_start[i1];
  }
}

__device__ __forceinline__ void pipe_finish(int lane, const PipeRaw& r, CellCtx& c) {
  c.cx = r.cx, c.cy = r.cy, c.cz = r.cz, c.wrap = 0x15;
  c.ibeg = __builtin_amdgcn_readfirstlane(r.ibeg);
  c.ni = __builtin_amdgcn_readfirstlane(r.iend) - c.ibeg;
  c.seg_src = r.seg_src;
  c.seg_len = r.seg_end - r.seg_src;
  c.seg_off = scan32_dpp(c.seg_len) - c.seg_len;
  c.total_j = __builtin_amdgcn_readlane(c.seg_off + c.seg_len, NSEG - 1);
}

// The stream of cell c into `dst`, asynchronously: wave v takes segments v, v + 8, v + 16.  Lane l's 16 bytes land at
// the (wave-uniform) LDS address + 16 l.  The sentinels up to the tile boundary are ordinary LDS writes.
template <int NW = PIPE_WAVES>
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

__device__ __forceinline__ void pipe_flush(const SweepArgs<float>& a, PipePending& p, int lane) {
  if (p.gcount <= 0) return;  // uniform
  // (the count first: its row index is the one load this waits for, and a wait placed behind the mask stores would be
  // a wait for them as well)
  if (lane < p.gcount) a.count[p.row_l] = p.mine;
#pragma unroll
  for (int k = 0; k < PIPE_G; k++)
    if (k < p.gcount) mask_store(a.masks, (size_t)(p.slot0 + k) * a.mask_nb, lane, p.w[k]);
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
template <bool FULL, int NW = PIPE_WAVES>
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
    pipe_flush(a, p, lane);  // (a wave with several groups: the previous one's words go out before the next search)
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

// Work distribution: the cells of the launch are cut into chunks of PIPE_CHUNK consecutive cells; XCD x owns the x-th
// eighth of the chunks (neighbouring cells share stencil cells: one L2), and the workgroups of an XCD draw chunks from
// its ticket counter (a.pipe_ticket[x], zeroed with the status word).  Dynamic on purpose: the instruction arbiter
// prefers the oldest waves of a SIMD, so of four persistent workgroups on a CU the first runs three times as fast as
// the last (tools/count_phases.py: 140 us against 414 us for the same 27 cells) -- with a fixed share per workgroup
// the launch lasts as long as its youngest workgroup.
// The pipeline looks two cells ahead, so the ticket of the chunk after the current one is drawn when the current one
// is entered (wave 0), handed to the other waves through LDS one barrier later and read one barrier after that:
// PIPE_CHUNK >= 4 cells separate the request from the first use.
constexpr int PIPE_CHUNK = 4;

template <bool FULL>
__global__ void __launch_bounds__(PIPE_WAVES* WAVE) __attribute__((amdgpu_num_sgpr(80))) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_sweep_pipe_f32(SweepArgs<float> a) {
  constexpr int CAP = PIPE_CAP;
  static_assert(PIPE_CAP % WAVE == 0 && PIPE_CAP <= SweepCfg<float>::CAP, "whole tiles; never more than the batch the expansion assumes");
  static_assert(PIPE_CHUNK >= 4, "the ticket of the next chunk takes two barriers to reach every wave");
  __shared__ __attribute__((aligned(32))) Pos<float> buf[2 * CAP];
  __shared__ int32_t s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if NL_PRIO
  __builtin_amdgcn_s_setprio(NL_PRIO);  // everything but the tile loop of search_group
#endif
  if (*a.status & ST_DOMAIN) return;  // (see cell_setup_at)
  const int32_t ncells = a.ncells_grid, nchunks = (ncells + PIPE_CHUNK - 1) / PIPE_CHUNK;
  const int32_t xcd = blockIdx.x & 7, cpx = (nchunks + 7) >> 3;  // chunks per XCD
  const int32_t chunk0 = xcd * cpx, chunk1 = min(chunk0 + cpx, nchunks);
  int32_t* const ticket = a.pipe_ticket + xcd;
  // ---- the sequence of cells of this workgroup
  int32_t pos = 0, pos_end = 0;  // rest of the current chunk
  int32_t next_chunk = -1;       // first cell of the chunk after it; -1: none (or not yet known: see PIPE_CHUNK)
  int32_t tk_phase = 0;          // 0 idle, 1 drawn (wave 0 holds it), 2 in LDS
  int32_t tk_val = 0;
  auto draw = [&]() {
    if (wave == 0 && lane == 0) tk_val = atomicAdd(ticket, 1);
    tk_phase = 1;
  };
  auto first_cell_of = [&](int32_t t) { return chunk0 + t < chunk1 ? (chunk0 + t) * PIPE_CHUNK : -1; };
  auto gen = [&]() -> int32_t {  // the next cell of the sequence, -1 when the XCD has no chunk left
    if (pos < pos_end) return pos++;
    if (next_chunk < 0) return -1;
    pos = next_chunk, pos_end = min(pos + PIPE_CHUNK, ncells), next_chunk = -1;
    draw();
    return pos++;
  };
  {  // the first chunk, synchronously
    if (wave == 0 && lane == 0) s_ticket = atomicAdd(ticket, 1);
    __syncthreads();
    next_chunk = first_cell_of(s_ticket);
    __syncthreads();
  }
  int32_t w_cur = gen();
  if (w_cur < 0) return;  // uniform
  PipeRaw raw;
  CellCtx cur, nxt;
  PipePending pend;
  pend.gcount = 0, pend.mine = 0, pend.row_l = 0, pend.slot0 = 0;
  // prologue: the first cell's table and stream, the second cell's table
  pipe_issue(a, lane, w_cur, true, raw);
  pipe_finish(lane, raw, cur);
  if (cur.ni > 0 && cur.total_j <= CAP) pipe_stage(a, cur, buf, tid, lane, wave);
  int32_t w_nxt = gen();
  pipe_issue(a, lane, max(w_nxt, 0), w_nxt >= 0, raw);
#if NL_STAMP
  uint64_t rt0;
  uint32_t hwid, xccid;
  asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)"
               : "=s"(rt0), "=s"(hwid), "=s"(xccid));
  uint64_t t_prev = __builtin_amdgcn_s_memtime(), t_acc[6] = {0, 0, 0, 0, 0, 0}, n_cells = 0;
  auto stamp = [&](int phase) {
    const uint64_t now = __builtin_amdgcn_s_memtime();
    t_acc[phase] += now - t_prev;
    t_prev = now;
  };
#endif
  int32_t parity = 0;
#pragma unroll 1
  while (w_cur >= 0) {
#if NL_STAMP
    stamp(0);  // loop overhead / prologue
#endif
    Pos<float>* const tcur = buf + parity * CAP;
    Pos<float>* const tnxt = buf + (parity ^ 1) * CAP;
    // the stream of `cur` has landed (own DMA: vmcnt; everybody's: barrier) and everybody has left `tnxt`
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#if NL_STAMP
    stamp(1);  // wait for the stream + barrier
#endif
    if (tk_phase == 2) {  // (uniform) the ticket wave 0 left in LDS one barrier ago
      next_chunk = first_cell_of(s_ticket);
      tk_phase = 0;
    } else if (tk_phase == 1) {  // the ticket drawn one barrier ago has arrived (vmcnt(0) above)
      if (wave == 0 && lane == 0) s_ticket = tk_val;
      tk_phase = 2;
    }
    pipe_flush(a, pend, lane);     // stores of the previous cell: a whole search away from the next wait
#if NL_STAMP
    stamp(2);  // deferred stores
#endif
    pipe_finish(lane, raw, nxt);   // table of the next cell (loaded during the previous step)
    const int32_t w_aft = w_nxt >= 0 ? gen() : -1;
    pipe_issue(a, lane, max(w_aft, 0), w_aft >= 0, raw);
#if NL_STAMP
    stamp(3);  // next table finished, the one after it issued
#endif
    if (nxt.ni > 0 && nxt.total_j <= CAP) pipe_stage(a, nxt, tnxt, tid, lane, wave);
#if NL_STAMP
    stamp(4);  // DMA issued
    n_cells++;
#endif
    if (cur.ni > 0) {
      if (cur.total_j <= CAP) {
        pipe_search<FULL>(a, cur, tcur, lane, wave, pend);
      } else if (tid == 0) {  // (rare) several LDS batches: k_sweep_list_f32
        a.full27_list[atomicAdd(a.full27_count, 1)] = cur.cx + (cur.cy + cur.cz * a.my) * a.mx;
      }
    }
    cur = nxt;
    w_cur = w_nxt, w_nxt = w_aft, parity ^= 1;
#if NL_STAMP
    stamp(5);  // search
#endif
  }
  pipe_flush(a, pend, lane);
#if NL_STAMP
  if (lane == 0) {
    unsigned long long* const slot = a.dbg_buf + 64 + (blockIdx.x & 1023) * 16;
    for (int ph = 0; ph < 6; ph++) atomicAdd(slot + ph, (unsigned long long)t_acc[ph]);
    atomicAdd(slot + 8, (unsigned long long)n_cells);
    atomicAdd(slot + 9, 1ull);
    if (wave == 0) {  // (last launch wins: one workgroup per slot when the grid has at most 1024 of them)
      uint64_t rt1;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1));
      slot[10] = rt0, slot[11] = rt1, slot[12] = ((unsigned long long)xccid << 32) | hwid;
    }
  }
#endif
}

// The same walk over chunks of cells with the workgroup shape of k_sweep_count_masks_f32: 4 waves, ONE buffer, 8
// workgroups per CU.  A cell's stream cannot overlap the previous cell's search here (two barriers per cell, the DMA
// round trip exposed), but what a workgroup per cell pays before its first useful instruction -- launch, kernel
// arguments, the segment table: two to three dependent round trips -- is gone: the table of the next cell is loaded
// during the search of this one.
constexpr int PERSIST_WAVES = SWEEP_WAVES;

template <bool FULL>
__global__ void __launch_bounds__(PERSIST_WAVES* WAVE, 8) __attribute__((amdgpu_num_sgpr(80)))
k_sweep_persist_f32(SweepArgs<float> a) {
  constexpr int CAP = PIPE_CAP, NW = PERSIST_WAVES;  // (19 KB + the ticket word: eight workgroups per CU)
  __shared__ __attribute__((aligned(32))) Pos<float> buf[CAP];
  __shared__ int32_t s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if NL_PRIO
  __builtin_amdgcn_s_setprio(NL_PRIO);  // everything but the tile loop of search_group
#endif
  if (*a.status & ST_DOMAIN) return;  // (see cell_setup_at)
  const int32_t ncells = a.ncells_grid, nchunks = (ncells + PIPE_CHUNK - 1) / PIPE_CHUNK;
  const int32_t xcd = blockIdx.x & 7, cpx = (nchunks + 7) >> 3;  // chunks per XCD
  const int32_t chunk0 = xcd * cpx, chunk1 = min(chunk0 + cpx, nchunks);
  int32_t* const ticket = a.pipe_ticket + xcd;
  int32_t pos = 0, pos_end = 0, next_chunk = -1, tk_phase = 0, tk_val = 0;  // (as k_sweep_pipe_f32)
  auto first_cell_of = [&](int32_t t) { return chunk0 + t < chunk1 ? (chunk0 + t) * PIPE_CHUNK : -1; };
  auto gen = [&]() -> int32_t {
    if (pos < pos_end) return pos++;
    if (next_chunk < 0) return -1;
    pos = next_chunk, pos_end = min(pos + PIPE_CHUNK, ncells), next_chunk = -1;
    if (wave == 0 && lane == 0) tk_val = atomicAdd(ticket, 1);
    tk_phase = 1;
    return pos++;
  };
  {
    if (wave == 0 && lane == 0) s_ticket = atomicAdd(ticket, 1);
    __syncthreads();
    next_chunk = first_cell_of(s_ticket);
    __syncthreads();
  }
  int32_t w_cur = gen();
  if (w_cur < 0) return;  // uniform
  PipeRaw raw;
  CellCtx cur;
  PipePending pend;
  pend.gcount = 0, pend.mine = 0, pend.row_l = 0, pend.slot0 = 0;
  pipe_issue(a, lane, w_cur, true, raw);
#pragma unroll 1
  while (w_cur >= 0) {
    pipe_finish(lane, raw, cur);  // this cell's table (loaded during the previous search)
    const int32_t w_nxt = gen();
    pipe_issue(a, lane, max(w_nxt, 0), w_nxt >= 0, raw);
    const bool single = cur.ni > 0 && cur.total_j <= CAP;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // everybody has left the buffer
    if (tk_phase == 2) {  // (uniform) the ticket wave 0 left in LDS one barrier ago
      next_chunk = first_cell_of(s_ticket);
      tk_phase = 0;
    }
    if (single) pipe_stage<NW>(a, cur, buf, tid, lane, wave);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // the stream has landed
    if (tk_phase == 1) {  // the ticket drawn before the wait above has arrived
      if (wave == 0 && lane == 0) s_ticket = tk_val;
      tk_phase = 2;
    }
    pipe_flush(a, pend, lane);  // stores of the previous cell
    if (single) {
      pipe_search<FULL, NW>(a, cur, buf, lane, wave, pend);
    } else if (cur.ni > 0 && tid == 0) {  // (rare) several LDS batches: k_sweep_list_f32
      a.full27_list[atomicAdd(a.full27_count, 1)] = cur.cx + (cur.cy + cur.cz * a.my) * a.mx;
    }
    w_cur = w_nxt;
  }
  pipe_flush(a, pend, lane);
}

// One workgroup per cell, like k_sweep_count_masks_f32, but only what a single-batch cell needs: the stream by LDS-DMA,
// the groups' i-particles taken from the staged stream (no loads of their own), no batch loop, no progress words; a
// cell whose stream does not fit goes on the list of k_sweep_list_f32.  (NL_PIPE=1.)
template <bool FULL>
__global__ void __launch_bounds__(SWEEP_WAVES* WAVE, 8) __attribute__((amdgpu_num_sgpr(80))) k_sweep_lean_f32(SweepArgs<float> a) {
  constexpr int CAP = SweepCfg<float>::CAP, NW = SWEEP_WAVES;
  __shared__ __attribute__((aligned(32))) Pos<float> buf[CAP];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#if NL_PRIO
  __builtin_amdgcn_s_setprio(NL_PRIO);  // everything but the tile loop of search_group
#endif
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
  pipe_flush(a, pend, lane);
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
