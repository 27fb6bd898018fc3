// nl_sweep_p.hpp -- persistent, software-pipelined pair search for fp32 positions (included by nl_kernels.hpp).
//
// Same arithmetic, same i-group / j-tile decomposition and the same search_group() inner loop as k_sweep, but
// organised around what the v2 profile showed (profiles/r01_*): waves spent most of their life waiting for the
// three dependent global round trips that stage a cell's stencil, so fewer than 5 of the 8 wave slots per SIMD
// were computing.  Here
//   * a workgroup of 8 waves stays resident and walks a contiguous range of i-cells (XCD-aware: each XCD gets a
//     contiguous z-slab of cells, each workgroup a contiguous run inside it, so consecutive cells re-use the
//     stencil data the previous cell pulled into L2/L1);
//   * the stencil of cell n+1 is copied global->LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, 1 KiB per wave
//     instruction, contiguous runs of the cell-sorted array map 1:1 onto the lane-linear LDS destination) into
//     the second LDS buffer WHILE cell n is being searched; the cell_start lookups behind that copy are issued
//     one more cell ahead, and the i-particles of cell n+1 (position, row / list offset) are prefetched too;
//   * one barrier per cell; 4 workgroups x 8 waves = 32 waves per CU, 2 x 20 KiB LDS each.
// A stencil longer than one LDS buffer (very dense cells) takes further, synchronous batches.
#pragma once

namespace nl {

constexpr int PW = 8;  // waves per workgroup

// Raw segment table of an i-cell: lane s < 18 loads the first and one-past-last cell_start entry of segment slot
// s, lane 18 does the same for the cell itself.  ONLY loads are issued here -- no arithmetic on the results -- so
// that they stay in flight until the next iteration picks them up (a use would make the compiler wait at once).
template <typename T>
__device__ __forceinline__ void load_cell_table(const SweepArgs<T>& a, int32_t cx, int32_t cy, int32_t cz, int lane,
                                                int32_t& lo, int32_t& hi) {
  lo = 0, hi = 0;
  if (lane <= NSEG) {
    int32_t i0, i1;
    if (lane == NSEG) {
      i0 = cx + (cy + cz * a.my) * a.mx, i1 = i0 + 1;
    } else {
      const int32_t s = lane % 9, part = lane / 9, dz = s / 3 - 1, dy = s % 3 - 1;
      int32_t y = cy + dy, z = cz + dz;
      if (y < 0) y += a.my;
      if (y >= a.my) y -= a.my;
      if (!a.slab) {
        if (z < 0) z += a.mzl;
        if (z >= a.mzl) z -= a.mzl;
      }
      int32_t x0, x1;
      if (cx == 0) {
        x0 = part ? 0 : a.mx - 1, x1 = part ? 2 : a.mx;
      } else if (cx == a.mx - 1) {
        x0 = part ? 0 : a.mx - 2, x1 = part ? 1 : a.mx;
      } else {
        x0 = cx - 1, x1 = part ? cx - 1 : cx + 2;
      }
      const int32_t rowbase = (y + z * a.my) * a.mx;
      i0 = rowbase + x0, i1 = rowbase + x1;
    }
    lo = a.cell_start[i0];
    hi = a.cell_start[i1];
  }
}

// LDS-DMA copy of the stream window [win0, win0 + CAP) of a cell's stencil into `buf`.  Wave w takes the segment
// slots w, w+8, w+16 (the nine never-empty slots 0..8 spread 2/1/1/1/1/1/1/1).
template <typename T, int CAP>
__device__ __forceinline__ void dma_stage(const SweepArgs<T>& a, Pos<T>* buf, int32_t seg_src, int32_t seg_len,
                                          int32_t seg_off, int32_t win0, int lane, int wave) {
  static_assert(sizeof(Pos<T>) == 16, "one particle = one 16-byte LDS-DMA element");
  for (int32_t sg = wave; sg < NSEG; sg += PW) {
    const int32_t len = __builtin_amdgcn_readlane(seg_len, sg);
    if (len == 0) continue;
    const int32_t src = __builtin_amdgcn_readlane(seg_src, sg);
    const int32_t off = __builtin_amdgcn_readlane(seg_off, sg) - win0;  // destination slot of element 0
    // clip to the window (uniform): elements [k0, k1) of this segment land in [0, CAP)
    const int32_t k0 = off < 0 ? -off : 0;
    const int32_t k1 = min(len, CAP - off);
    for (int32_t kb = k0; kb < k1; kb += WAVE) {
      const int32_t k = kb + lane;
      if (k < k1) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(a.sorted + src + k),
            (__attribute__((address_space(3))) void*)(buf + off + kb), 16, 0, 0);
      }
    }
  }
}

template <typename T, int MODE>
__global__ void __launch_bounds__(PW* WAVE, 8) __attribute__((amdgpu_num_sgpr(80)))
k_sweep_p(SweepArgs<T> a, int32_t ncells_i, const int32_t* __restrict__ base_sorted, int32_t* __restrict__ tickets) {
  constexpr int CAP = SweepCfg<T>::CAP - WAVE;  // 1216: 2 buffers + ticket slots = 38 KiB, four workgroups per CU
  constexpr int G = SWEEP_G;
  // one LDS array (a second __shared__ object next to an LDS-DMA target makes hipcc drain the DMA before every
  // ds_read): two stencil buffers + two spare slots whose gid fields carry the cell tickets to all waves
  __shared__ Pos<T> lds[2 * CAP + 2];
  Pos<T>* const tile0 = lds;
  Pos<T>* const tile1 = lds + CAP;
  int32_t* const s_ticket = &lds[2 * CAP].gid;  // slot cb: [2*CAP + cb].gid (two slots: a write never races a late read)

  if (MODE == MODE_FILL) {
    if (a.total[0] > a.capacity) {
      if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(a.status, ST_CAPACITY);
      return;
    }
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // Work distribution: the i-cells are cut into 8 contiguous ranges (z-slabs), one per XCD -- blocks b, b+8, ...
  // share an XCD and its L2 -- and the workgroups of an XCD draw cells from their range one at a time through an
  // atomic ticket.  (A static split left the four workgroups of a CU finishing at 190/270/360/445 us: the SIMD
  // arbitrates oldest-wave-first, so the youngest workgroup ran its tail alone at a quarter of the occupancy.)
  const int32_t xcd = blockIdx.x & 7;
  const int32_t xb = (int32_t)((int64_t)ncells_i * xcd / 8), xlen = (int32_t)((int64_t)ncells_i * (xcd + 1) / 8) - xb;
  int32_t* const my_ticket = tickets + xcd;
  unsigned long long t_wg_start = 0;
  if (a.dbg & 8) asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_wg_start)::"memory");

  auto cell_of_ticket = [&](int32_t t) { return t < xlen ? xb + t : -1; };
  auto coords = [&](int32_t w, int32_t& cx, int32_t& cy, int32_t& cz) {
    const int32_t wy = (int32_t)fastdiv((uint32_t)w, a.div_mx), wz = (int32_t)fastdiv((uint32_t)wy, a.div_my);
    cx = w - wy * a.mx, cy = wy - wz * a.my, cz = wz + (a.slab ? 1 : 0);
  };
  auto group_shape = [&](int32_t n_i, int32_t& rounds, int32_t& gsize) {
    rounds = (n_i + PW * G - 1) / (PW * G);
    // one round (cells of up to 48 particles) needs only a shift; denser cells pay a real division
    gsize = rounds <= 1 ? (n_i + PW - 1) / PW : (n_i + rounds * PW - 1) / (rounds * PW);
  };
  // i-particles of this wave's FIRST group of a cell (the only group unless the cell holds more than 48)
  auto load_group = [&](int32_t ib, int32_t n_i, int32_t g, int32_t gsize, Pos<T>& pi_l, int32_t& row_l, int32_t& base_l) {
    const int32_t i0 = g * gsize, gcount = min(gsize, n_i - i0);
    pi_l.x = 0, pi_l.y = 0, pi_l.z = 0, pi_l.gid = 0;
    row_l = 0, base_l = 0;
    if (lane < gcount) {
      pi_l = a.sorted[ib + i0 + lane];
      if (MODE == MODE_FILL) base_l = base_sorted[ib + i0 + lane];
      else row_l = a.sorted_row[ib + i0 + lane];
    }
  };

  // ---- pipeline prologue: three tickets (cells n, n+1, n+2), tables of the first two
  int32_t tk_pending = 0;  // thread 0: ticket whose atomic is in flight, written to LDS before the next barrier
  if (tid == 0) {
    const int32_t t0 = atomicAdd(my_ticket, 3);
    *s_ticket = t0;
  }
  __syncthreads();
  const int32_t t_first = *s_ticket;
  int32_t w_cur = cell_of_ticket(t_first);      // cell n
  int32_t w_nxt = cell_of_ticket(t_first + 1);  // cell n+1
  int32_t w_nn = cell_of_ticket(t_first + 2);   // cell n+2 (its table is loaded in iteration n)
  if (w_cur < 0) {
    if ((a.dbg & 8) && tid == 0) {
      a.dbg_buf[64 + 4 * blockIdx.x + 0] = t_wg_start;
      a.dbg_buf[64 + 4 * blockIdx.x + 1] = t_wg_start + 1;
      a.dbg_buf[64 + 4 * blockIdx.x + 3] = 0;
    }
    return;
  }
  if (tid == 0) tk_pending = atomicAdd(my_ticket, 1);  // cell n+3

  int32_t ibeg, ni, seg_src, seg_len, seg_off, total_j;  // cell n, finalised
  int32_t lo1 = 0, hi1 = 0, lo2 = 0, hi2 = 0;            // raw tables of cells n+1 and n+2 (lane 18 = the cell itself)
  {
    int32_t cx, cy, cz, lo0, hi0;
    coords(w_cur, cx, cy, cz);
    load_cell_table(a, cx, cy, cz, lane, lo0, hi0);
    if (w_nxt >= 0) {
      coords(w_nxt, cx, cy, cz);
      load_cell_table(a, cx, cy, cz, lane, lo1, hi1);
    }
    ibeg = __builtin_amdgcn_readlane(lo0, NSEG);
    ni = __builtin_amdgcn_readlane(hi0, NSEG) - ibeg;
    seg_src = lo0, seg_len = lane < NSEG ? hi0 - lo0 : 0;
    seg_off = scan32_dpp(seg_len) - seg_len;
    total_j = __builtin_amdgcn_readlane(seg_off + seg_len, NSEG - 1);
  }
  int32_t rounds, gsize;
  group_shape(ni, rounds, gsize);
  Pos<T> pi_l;
  int32_t row_l, base_l;
  load_group(ibeg, ni, wave, gsize, pi_l, row_l, base_l);
  if (ni > 0) dma_stage<T, CAP>(a, tile0, seg_src, seg_len, seg_off, 0, lane, wave);

  // diagnostics (dbg & 4): shader cycles this wave spends waiting at the barrier / in the look-ahead / searching
  unsigned long long t_bar = 0, t_look = 0, t_search = 0, t_prev = 0, t_other = 0;
  auto stamp = [&](unsigned long long& acc) {
    if (a.dbg & 4) {
      unsigned long long t;
      asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
      acc += t - t_prev;
      t_prev = t;
    }
  };
  if (a.dbg & 4) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t_prev)::"memory");
  int32_t ncell_done = 0;

  for (int cb = 0; w_cur >= 0; cb ^= 1) {
    Pos<T>* cur = cb ? tile1 : tile0;
    Pos<T>* nxt = cb ? tile0 : tile1;
    const int32_t nj0 = min(total_j, CAP);
    // sentinel padding of the current buffer (its DMA was issued one iteration ago; distinct addresses)
    if (ni > 0) {
      const int32_t pad = nj0 + tid;
      if (pad < ((nj0 + WAVE - 1) & ~(WAVE - 1))) {
        Pos<T> sentinel;
        sentinel.x = 0, sentinel.y = 0, sentinel.z = 0, sentinel.gid = INT32_MIN;
        cur[pad] = sentinel;
      }
    }
    int32_t* const tk_slot = &lds[2 * CAP + cb].gid;
    if (tid == 0) *tk_slot = tk_pending;  // the ticket drawn one iteration ago (cell n+3)
    stamp(t_other);
    __syncthreads();  // (waits vmcnt(0): the DMA of cell n and every prefetched value have landed)
    stamp(t_bar);
    const int32_t w_n3 = cell_of_ticket(*tk_slot);
    if (tid == 0 && w_n3 >= 0) tk_pending = atomicAdd(my_ticket, 1);  // cell n+4; result used next iteration

    // ---- look ahead.  First finalise the table of cell n+1 (its loads were issued one iteration ago) ...
    const int32_t ibeg1 = __builtin_amdgcn_readlane(lo1, NSEG);
    const int32_t ni1 = __builtin_amdgcn_readlane(hi1, NSEG) - ibeg1;
    const int32_t seg_src1 = lo1, seg_len1 = lane < NSEG ? hi1 - lo1 : 0;
    const int32_t seg_off1 = scan32_dpp(seg_len1) - seg_len1;
    const int32_t total_j1 = __builtin_amdgcn_readlane(seg_off1 + seg_len1, NSEG - 1);
    int32_t rounds1, gsize1;
    group_shape(ni1, rounds1, gsize1);
    // ... then ISSUE (and only issue) everything the next iterations need: raw table of cell n+2, the i-particles
    // of cell n+1, and the LDS-DMA copy of cell n+1's stencil into the other buffer.
    lo2 = 0, hi2 = 0;
    if (w_nn >= 0) {
      int32_t cx, cy, cz;
      coords(w_nn, cx, cy, cz);
      load_cell_table(a, cx, cy, cz, lane, lo2, hi2);
    }
    Pos<T> pi_n;
    int32_t row_n, base_n;
    load_group(ibeg1, ni1, wave, gsize1, pi_n, row_n, base_n);
    if (w_nxt >= 0 && ni1 > 0 && !(a.dbg & 2)) dma_stage<T, CAP>(a, nxt, seg_src1, seg_len1, seg_off1, 0, lane, wave);

    stamp(t_look);
    // ---- search cell n
    if (ni > 0 && !(a.dbg & 1)) {
      const int32_t nbatch = (total_j + CAP - 1) / CAP;
      for (int32_t batch = 0; batch < nbatch; batch++) {
        const int32_t win0 = batch * CAP;
        const int32_t nj = min(total_j - win0, CAP);
        if (batch) {  // rare: stencil longer than one buffer -> further batches, staged synchronously into `cur`
          __syncthreads();
          dma_stage<T, CAP>(a, cur, seg_src, seg_len, seg_off, win0, lane, wave);
          const int32_t pad = nj + tid;
          if (pad < ((nj + WAVE - 1) & ~(WAVE - 1))) {
            Pos<T> sentinel;
            sentinel.x = 0, sentinel.y = 0, sentinel.z = 0, sentinel.gid = INT32_MIN;
            cur[pad] = sentinel;
          }
          __syncthreads();
        }
        const int32_t ntiles = (nj + WAVE - 1) / WAVE;
        // one group of this wave: pg / rowg / baseg are its i-particles (lane k < gcount)
        auto do_group = [&](int32_t g, const Pos<T>& pg, int32_t rowg, int32_t baseg) {
          const int32_t i0 = g * gsize;
          const int32_t gcount = min(gsize, ni - i0);
          if (gcount <= 0) return;
          if (MODE == MODE_FILL && batch) {
            // the row of a FILL lane is only needed for the progress scratch of multi-batch cells
            rowg = lane < gcount ? a.sorted_row[ibeg + i0 + lane] : 0;
          }
          if (batch && lane < gcount) baseg += a.progress[rowg];
          int32_t mine;
          switch (gcount) {
            case 1: mine = search_group<T, MODE, 1>(a, cur, nj, ntiles, lane, pg, baseg); break;
            case 2: mine = search_group<T, MODE, 2>(a, cur, nj, ntiles, lane, pg, baseg); break;
            case 3: mine = search_group<T, MODE, 3>(a, cur, nj, ntiles, lane, pg, baseg); break;
            case 4: mine = search_group<T, MODE, 4>(a, cur, nj, ntiles, lane, pg, baseg); break;
            default: mine = search_group<T, MODE, 5>(a, cur, nj, ntiles, lane, pg, baseg); break;
          }
          if (lane < gcount) {
            if (nbatch > 1) {
              if (MODE == MODE_FILL && !batch) rowg = a.sorted_row[ibeg + i0 + lane];
              const int32_t before = batch ? a.progress[rowg] : 0;
              mine += before;
              a.progress[rowg] = mine;
            }
            if (MODE == MODE_COUNT && batch == nbatch - 1) a.count[rowg] = mine;
          }
        };
        // first group: its particles were prefetched one cell ahead (no load, hence no wait, in this path)
        do_group(wave, pi_l, row_l, base_l);
        // cells with more than PW*G particles: further groups, loaded on the spot
        for (int32_t r = 1; r < rounds; r++) {
          Pos<T> pg;
          int32_t rowg, baseg;
          load_group(ibeg, ni, wave + r * PW, gsize, pg, rowg, baseg);
          do_group(wave + r * PW, pg, rowg, baseg);
        }
      }
    }

    stamp(t_search);
    // ---- rotate the pipeline registers
    ibeg = ibeg1, ni = ni1, seg_src = seg_src1, seg_len = seg_len1, seg_off = seg_off1, total_j = total_j1;
    rounds = rounds1, gsize = gsize1;
    pi_l = pi_n, row_l = row_n, base_l = base_n;
    lo1 = lo2, hi1 = hi2;
    w_cur = w_nxt, w_nxt = w_nn, w_nn = w_n3;
    ncell_done++;
  }
  if ((a.dbg & 8) && tid == 0) {  // per-workgroup record: start, end (s_memrealtime, 100 MHz), HW ids
    unsigned long long t_end;
    unsigned int hwid, xcc;
    asm volatile("s_memrealtime %0\n s_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n s_waitcnt lgkmcnt(0)"
                 : "=s"(t_end), "=s"(hwid), "=s"(xcc)::"memory");
    a.dbg_buf[64 + 4 * blockIdx.x + 0] = t_wg_start;
    a.dbg_buf[64 + 4 * blockIdx.x + 1] = t_end;
    a.dbg_buf[64 + 4 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hwid;
    a.dbg_buf[64 + 4 * blockIdx.x + 3] = (unsigned long long)ncell_done;
  }
  if ((a.dbg & 4) && lane == 0) {
    atomicAdd(a.dbg_buf + 0, t_bar);
    atomicAdd(a.dbg_buf + 1, t_look);
    atomicAdd(a.dbg_buf + 2, t_search);
    atomicAdd(a.dbg_buf + 3, t_other);
    atomicAdd(a.dbg_buf + 4, 1ull);
  }
}

// base_sorted[slot] = key_pointer[sorted_row[slot]]: the list offset of every row, in cell order, so that the FILL
// pass needs one (prefetchable) load per i-particle instead of two dependent ones.
__global__ void __launch_bounds__(256) k_row_base(const int32_t* __restrict__ key_pointer,
                                                   const int32_t* __restrict__ sorted_row, int32_t n_rows, int32_t n,
                                                   int32_t* __restrict__ base_sorted) {
  const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int32_t r = sorted_row[s];
  // ghosts (slab builds) have no row; in a build that failed its checks a slot may never have been written: unsigned
  // compare, so that whatever it holds is not used as an index
  base_sorted[s] = (uint32_t)r < (uint32_t)n_rows ? key_pointer[r] : 0;
}

}  // namespace nl
