// nl_api.hip -- host side of libnl_hip.so: the C ABI of include/nl_hip.h over the kernels of nl_kernels.hpp.
// No torch, no CUDA-compat layer: HIP runtime only.  Compiled for gfx950 with -ffp-contract=off.
#include "../../include/nl_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "nl_kernels.hpp"

using namespace nl;

namespace {

// pinned; mirrors the int32 words [status, 16 tickets, pad, total_lo, total_hi] that follow the cell histogram on the
// device; filled by ONE async D2H copy at the end of every build
struct HostResult {
  uint32_t status;
  uint32_t tickets[17];
  uint32_t total_lo, total_hi;
  int64_t total() const { return (int64_t)(((uint64_t)total_hi << 32) | total_lo); }
};
constexpr int META_TOTAL = 18;  // int32 offset of total_lo from the status word
constexpr int META_FULL27 = 1;  // number of cells the COUNT sweep hands to the batched search (first "ticket" word)
constexpr int META_FILL_LIST = 10;  // number of cells k_fill_masks hands to k_fill_list ("ticket" word 10)
constexpr int META_WORDS = 20;

}  // namespace

struct nl_handle_s {
  int dtype = NL_F32, device = 0;
  double rc = 0, L[3] = {0, 0, 0}, rc2 = 0;
  int32_t m[3] = {0, 0, 0};
  int64_t ncell = 0;
  float ims_f[3];
  double ims_d[3];
  float rc2_f = 0;
  float ms_f[3];       // cell edge as the reference's float Vec holds it (neighlist_cpu.hpp:389-391)

  int32_t n_max = 0;
  int64_t capacity = 0;      // list entries (half pairs, or twice as many for a full list)
  bool capacity_user = false;
  int list_kind = NL_LIST_HALF;
  bool pbc = false;          // minimum-image mode (nl_set_periodic)

  // device buffers
  int32_t* rank = nullptr;
  void* sorted = nullptr;
  int32_t* sorted_row = nullptr;
  int32_t* sorted_gid = nullptr;   // ids in cell order, compact
  int32_t* count = nullptr;
  void* key_pointer = nullptr;     // [n_rows + 1] int32 (the reference's type, neighlist_cpu.hpp:29) or, in a wide build, int64
  void* kp_alt = nullptr;          // key_pointer converted to the other width on demand (nl_get_*_csr / nl_get_*_csr64)
  bool kp_alt_valid = false;
  int offset_width = 0;            // nl_set_offset_width: 0 = by capacity (int64 as soon as the list may exceed INT32_MAX), 32, 64
  int32_t* progress = nullptr;
  // two-level binning (k_bin_*)
  int32_t* row_count = nullptr;   // [nrows] zeroed per build, then row totals
  int32_t* row_start = nullptr;   // [nrows + 1]
  int32_t* blk_base = nullptr;    // [bin_blocks][nrows]
  void* tmp_pos = nullptr;        // particles grouped by row
  int32_t* tmp_row = nullptr;
  int32_t bin_blocks = 0, bin_chunk = 0;
  bool bin_two_level = true;      // NL_BINNING=1 selects the atomic-rank path (k_hash/k_reorder)
  void* base_sorted = nullptr;     // key_pointer of every sorted slot (mask expansion), same width as key_pointer
  uint32_t* masks = nullptr;       // [n][64] hit bits of every sorted slot, between COUNT_MASKS and k_fill_masks
  int32_t* full27_list = nullptr;
  void* resort_buf = nullptr;      // scratch of nl_resort (32 bytes per particle), allocated on first use
  int32_t b_mask_nb = 1;           // this build: mask rows per sorted slot (> 1: dense build, k_fill_dense)
  size_t masks_bytes = 0;          // size of the masks allocation
  int32_t b_isplit = 1;            // this build, two-sweep path: workgroups per cell
  int isplit_env = 0;              // NL_ISPLIT: 0 = by density
  int rows_env = -1;               // NL_ROWS: -1 (default) = the fine-row search where the 27-cell path would need several LDS batches
                                   // per cell (denser than 40.3 particles per cell), 0 = never, 1..3 = RowsCfg<V - 1> wherever a
                                   // build qualifies (tests), 4 = wherever a build qualifies, RowsCfg by density (sweeps)
  bool b_lean_small = false;       // this build: the 2-wave, half-buffer instance of k_sweep_lean_f32 (sparse boxes)
  int lean_small_env = 1;          // NL_LEAN_SMALL=0: never (same-box A/B)
  int fill_small_env = 1;          // NL_FILL_SMALL=0: the 2-wave expansion also in sparse boxes (same-box A/B)
  bool b_rows = false;             // this build: fine rows (k_bin_cells<FINE>, k_sweep_rows_f32, k_fill_rows); the cell table is
  int b_rows_v = 0;                // fine_start (4 M + 1 entries); RowsCfg of the build
  bool dense_masks_off = false;    // NL_DENSE_MASKS=0: dense builds use two distance sweeps (the round-1 path)
  size_t dense_masks_limit = (size_t)64 << 30;  // most memory the mask rows of a dense build may take
  int sweep_variant = 3;           // 1: COUNT + FILL distance sweeps;
                                   // 3 (default): COUNT keeping hit masks + mask expansion
                                   // (2 = persistent LDS-DMA sweeps, 4 / 5 = matrix-core searches: measured slower or a draw
                                   // in round 1 and removed)
  int num_cus = 256;
  unsigned long long* dbg_buf = nullptr;
  int dbg_flags = 0, dbg_wg_per_cu = 4;  // diagnostics (NL_DEBUG_FLAGS, NL_DEBUG_WG_PER_CU)
  int dbg_lds_pad = 0;                   // diagnostics (NL_DEBUG_LDS_PAD): extra dynamic LDS bytes on the COUNT_MASKS
                                         // launch = fewer resident workgroups per CU (occupancy experiments)
  int32_t* cell_count = nullptr;  // [ncell] followed by the status word
  int32_t* cell_start = nullptr;  // [ncell + 1]
  uint64_t* scan_look = nullptr;  // k_scan_chained: [scan_blocks] entries + the two counters; all zero between launches
  int32_t scan_blocks = 0;
  int64_t* totals = nullptr;  // [0] = particles (cell scan), [1] = pairs (row scan)
  uint32_t* status = nullptr;
  int32_t* list = nullptr;
  // transposed full list (compat output)
  int32_t* t_list = nullptr;
  int32_t* t_count = nullptr;
  int32_t* t_cursor = nullptr;
  int64_t t_rows_cap = 0;
  int32_t t_max = 0;
  bool t_valid = false;

  HostResult* host = nullptr;
  hipStream_t own_stream = nullptr;
  // NL_GRAPH=1 / nl_set_graph: asynchronous builds are replayed from a captured hipGraph (one graph per argument set;
  // re-captured when an argument or any buffer changes).  Saves launch overhead on small systems.
  bool use_graph = false;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    const void* q = nullptr;
    const int32_t* gid = nullptr;
    int32_t stride = 0, n_rows = 0, n = 0, z_lo = 0, mzl = 0, slab = 0, list_kind = 0, pbc = 0, offset_width = 0;
    int64_t capacity = 0;
    uint64_t epoch = 0;
    bool operator==(const GraphKey& o) const {
      return q == o.q && gid == o.gid && stride == o.stride && n_rows == o.n_rows && n == o.n && z_lo == o.z_lo &&
             mzl == o.mzl && slab == o.slab && list_kind == o.list_kind && pbc == o.pbc && offset_width == o.offset_width &&
             capacity == o.capacity &&
             epoch == o.epoch;
    }
  } graph_key;
  uint64_t buffers_epoch = 1;  // bumped by every (re)allocation
  bool begun = false;  // nl_make_list_slab_begin has run, nl_make_list_slab_finish has not
  int32_t begun_ghost_lo = 0, begun_zlo = 0, begun_zhi = 0;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev[NL_NUM_STAGES + 1] = {};

  // state of the last build
  bool built = false, pending = false;
  int32_t n = 0, n_rows = 0;
  int64_t ncell_local = 0;
  int last_error = NL_OK, last_hip = 0;
  // arguments of the last build (to re-run the fill after growing the list)
  int32_t b_mzl = 0, b_slab = 0, b_zlo = 0, b_stride = 4;
  bool b_use_masks = false;  // this build: COUNT keeps hit masks and the list is expanded from them
  bool b_wide = false;       // this build: key_pointer / base_sorted hold int64 (the list may exceed INT32_MAX entries)
  bool b_full = false;       // this build: full list (both directions), nl_set_list_kind
  bool b_pbc = false;        // this build: minimum-image distances (nl_set_periodic)
  int b_variant = 3;         // sweep variant of this build (a full build uses 1 or 3 only)
  const void* b_q = nullptr;
  const int32_t* b_gid = nullptr;
  // nl_make_list_distributed: the ghost counts of the build live on the device (b_dyn[0], b_dyn[1]; n is an upper bound);
  // dyn_host: where their pinned copy (words 2, 3) and the exchange's error flags (word 4) arrive with the build's result
  const int32_t* b_dyn = nullptr;
  int32_t b_n_est = 0;             // particles expected (owned + the previous build's ghosts): path selection only
  const int32_t* dyn_host = nullptr;
};

namespace {

#define HIPCHK(h, call)                         \
  do {                                          \
    hipError_t e_ = (call);                     \
    if (e_ != hipSuccess) {                     \
      (h)->last_hip = (int)e_;                  \
      (h)->last_error = NL_ERR_HIP;             \
      return NL_ERR_HIP;                        \
    }                                           \
  } while (0)

int fail(nl_handle_t h, int code) {
  if (h) h->last_error = code;
  return code;
}

template <typename P> int dev_alloc(nl_handle_t h, P** p, size_t bytes) {
  h->buffers_epoch++;  // a captured graph holds the old pointers
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(p), bytes ? bytes : 16);
  if (e != hipSuccess) {
    h->last_hip = (int)e;
    return fail(h, e == hipErrorOutOfMemory ? NL_ERR_NOMEM : NL_ERR_HIP);
  }
  return NL_OK;
}

float floor_to_float(double v) {  // largest float <= v
  float f = (float)v;
  if ((double)f > v) f = std::nextafterf(f, -INFINITY);
  return f;
}

int status_to_error(uint32_t st) {
  if (st & ST_OUT_OF_BOX) return NL_ERR_OUT_OF_BOX;
  if (st & ST_DOMAIN) return NL_ERR_DOMAIN;
  if (st & ST_INDEX_OVERFLOW) return NL_ERR_INDEX_OVERFLOW;
  if (st & ST_CAPACITY) return NL_ERR_CAPACITY;
  return NL_OK;
}

template <typename T> Grid<T> make_grid(nl_handle_t h, int32_t n_rows, int32_t z_lo, int32_t mzl, int32_t slab) {
  Grid<T> g;
  for (int d = 0; d < 3; d++) {
    g.ims[d] = sizeof(T) == 4 ? (T)h->ims_f[d] : (T)h->ims_d[d];
    g.m[d] = h->m[d];
  }
  g.mzl = mzl;
  g.slab = slab;
  g.z_origin = slab ? ((z_lo - 1) % h->m[2] + h->m[2]) % h->m[2] : 0;
  g.n_rows = n_rows;
  g.pbc = h->pbc ? 1 : 0;
  g.dbg = h->dbg_flags;
  g.z_first = slab ? z_lo - 1 : 0;
  for (int d = 0; d < 3; d++) g.L[d] = (T)h->L[d];
  return g;
}

// exclusive scan of in[n] into out[n+1]; grand total to total[0]
template <typename OFF>
int launch_scan(nl_handle_t h, const int32_t* in, int64_t n, OFF* out, int64_t* total, hipStream_t s,
                uint32_t* total_split = nullptr) {
  if (n <= 0) {  // nothing to scan: out[0] = 0, total = 0 (a zero-size grid is not a valid launch)
    HIPCHK(h, hipMemsetAsync(out, 0, sizeof(OFF), s));
    HIPCHK(h, hipMemsetAsync(total, 0, sizeof(int64_t), s));
    if (total_split) HIPCHK(h, hipMemsetAsync(total_split, 0, 2 * sizeof(uint32_t), s));
    return NL_OK;
  }
  if (n <= SCAN_SMALL_MAX) {  // one launch instead of three (totals of such short arrays fit int32)
    hipLaunchKernelGGL(k_scan_small<OFF>, dim3(1), dim3(1024), 0, s, in, (int32_t)n, total, out, total_split);
    return NL_OK;
  }
  const int32_t nb = (int32_t)((n + SCAN_BLOCK - 1) / SCAN_BLOCK);
  if (nb > h->scan_blocks) return fail(h, NL_ERR_ARG);  // (sized for max(n_max, cells) in nl_reserve)
  hipLaunchKernelGGL(k_scan_chained<OFF>, dim3(nb), dim3(SCAN_THREADS), 0, s, in, n, h->scan_look, h->scan_blocks, total, out,
                     h->status, total_split);
  return NL_OK;
}

FastDiv fastdiv_make(uint32_t d) {  // see fastdiv() in nl_kernels.hpp; valid for dividends below 2^31
  FastDiv f;
  f.d = d;
  uint32_t s = 0;
  while ((1ull << s) < d) s++;
  f.s = s;
  f.m = (uint32_t)(((1ull << 32) * ((1ull << s) - d)) / d + 1);
  return f;
}

template <typename T> SweepArgs<T> sweep_args(nl_handle_t h) {
  SweepArgs<T> a;
  a.sorted = static_cast<const Pos<T>*>(h->sorted);
  a.sorted_row = h->sorted_row;
  a.sorted_gid = h->sorted_gid;
  a.cell_start = h->cell_start;
  a.mx = h->m[0], a.my = h->m[1], a.mzl = h->b_mzl, a.slab = h->b_slab;
  a.div_mx = fastdiv_make((uint32_t)h->m[0]), a.div_my = fastdiv_make((uint32_t)h->m[1]);
  a.rc2 = sizeof(T) == 4 ? (T)h->rc2_f : (T)h->rc2;
  a.count = h->count;
  a.progress = h->progress;
  a.key_pointer = h->key_pointer;
  a.wide = h->b_wide ? 1 : 0;
  a.n_rows = h->n_rows;
  a.list = h->list;
  a.total = h->totals + 1;
  a.capacity = h->capacity;
  a.status = h->status;
  a.masks = h->masks;
  // (the two planes of the hit words, one array each inside the same allocation: 128 + 64 bytes per row)
  a.masks_hi = reinterpret_cast<uint8_t*>(h->masks) + h->masks_bytes / MASK_ROW_BYTES * MASK_LO_BYTES;
  a.isplit = 1;
  a.mask_nb = h->b_mask_nb;
  a.full27_list = h->full27_list;
  a.full27_count = reinterpret_cast<int32_t*>(h->status) + META_FULL27;
  a.fill_list_count = reinterpret_cast<int32_t*>(h->status) + META_FILL_LIST;
  a.pbc = h->pbc ? 1 : 0;
  for (int d = 0; d < 3; d++) a.ms[d] = (T)(h->L[d] / h->m[d]);
  for (int d = 0; d < 3; d++) a.L[d] = (T)h->L[d];
  a.z_origin = h->b_slab ? h->b_zlo - 1 : 0;
  a.dbg = h->dbg_flags;
  a.dbg_buf = h->dbg_buf;
  return a;
}

RowsArgs rows_args(nl_handle_t h) {
  RowsArgs a;
  a.sorted = static_cast<const Pos<float>*>(h->sorted);
  a.sorted_row = h->sorted_row, a.sorted_gid = h->sorted_gid;
  a.fine_start = h->cell_start;
  a.mx = h->m[0], a.my = h->m[1], a.mzl = h->b_mzl, a.slab = h->b_slab;
  a.div_mx = fastdiv_make((uint32_t)h->m[0]), a.div_my = fastdiv_make((uint32_t)h->m[1]);
  a.rc2 = h->rc2_f;
  a.count = h->count;
  a.masks = h->masks;
  a.key_pointer = h->key_pointer;
  a.list = h->list;
  a.total = h->totals + 1;
  a.capacity = h->capacity;
  a.status = h->status;
  a.over_list = h->full27_list;
  a.over_count = reinterpret_cast<int32_t*>(h->status) + META_FULL27;
  a.wide = h->b_wide ? 1 : 0;
  a.dbg_buf = h->dbg_buf;
  return a;
}

// The fine-row path (nl_rows.hpp).  V: RowsCfg of the build.
template <int V, bool FULL> void launch_rows(nl_handle_t h, int mode, int32_t ncells_i, hipStream_t s) {
  const RowsArgs a = rows_args(h);
  const int32_t over_grid = 2 * h->num_cus;
  if (mode == MODE_COUNT) {
    hipLaunchKernelGGL((k_sweep_rows_f32<V, FULL>), dim3(ncells_i), dim3(ROWS_WAVES * WAVE), 0, s, a);
    hipLaunchKernelGGL((k_rows_overflow<MODE_COUNT, FULL, int32_t>), dim3(over_grid), dim3(ROWS_WAVES * WAVE), 0, s, a);
    return;
  }
  if (h->b_wide) {
    hipLaunchKernelGGL((k_fill_rows<V, FULL, int64_t>), dim3(ncells_i), dim3(ROWS_FW * WAVE), 0, s, a);
    hipLaunchKernelGGL((k_rows_overflow<MODE_FILL, FULL, int64_t>), dim3(over_grid), dim3(ROWS_WAVES * WAVE), 0, s, a);
  } else {
    hipLaunchKernelGGL((k_fill_rows<V, FULL, int32_t>), dim3(ncells_i), dim3(ROWS_FW * WAVE), 0, s, a);
    hipLaunchKernelGGL((k_rows_overflow<MODE_FILL, FULL, int32_t>), dim3(over_grid), dim3(ROWS_WAVES * WAVE), 0, s, a);
  }
}

// FULL = the list keeps both directions of every pair (the reference GPU class's contract).
template <typename T, bool FULL, bool PBC, typename OFF> void launch_fill_masks(nl_handle_t h, const SweepArgs<T>& a, int32_t ncells_i, hipStream_t s) {
  if (h->b_mask_nb > 1) {  // dense build: mask rows per (slot, LDS batch); list offsets gathered into cell order first
    const int32_t nbp = (h->n + 255) / 256;
    if (h->n > 0)
      hipLaunchKernelGGL(k_row_base<OFF>, dim3(nbp), dim3(256), 0, s, static_cast<const OFF*>(h->key_pointer), h->sorted_row,
                         h->n_rows, h->n, static_cast<OFF*>(h->base_sorted));
    hipLaunchKernelGGL((k_fill_dense<T, FULL, PBC, OFF>), dim3(ncells_i), dim3(FD_WAVES * WAVE), 0, s, a,
                       static_cast<const OFF*>(h->base_sorted));
    return;
  }
  // rows a wave loads up front: 24, or 12 where cells hold ~20 particles or fewer (a wave then has ~10 rows)
  const bool few_rows = (double)h->n <= 21.0 * (double)std::max<int64_t>(1, h->ncell_local);
  if constexpr (sizeof(T) == 4 && !PBC) {
    if (h->b_lean_small && h->fill_small_env) {  // sparse boxes: a wave per cell (its ~19 rows in one batch), ids of half a stream
      hipLaunchKernelGGL((k_fill_masks<T, FULL, PBC, OFF, 24, 1, SweepCfg<T>::CAP / 2>), dim3(ncells_i), dim3(WAVE), 0, s, a);
      hipLaunchKernelGGL((k_fill_list<T, FULL, PBC>), dim3(2 * h->num_cus), dim3(SWEEP_WAVES * WAVE), 0, s, a);
      return;
    }
  }
  if (few_rows)
    hipLaunchKernelGGL((k_fill_masks<T, FULL, PBC, OFF, 12>), dim3(ncells_i), dim3(EXPAND_WAVES * WAVE), 0, s, a);
  else
    hipLaunchKernelGGL((k_fill_masks<T, FULL, PBC, OFF, 24>), dim3(ncells_i), dim3(EXPAND_WAVES * WAVE), 0, s, a);
  // cells without masks (a stream of several LDS batches among one-batch neighbours): a second distance search
  hipLaunchKernelGGL((k_fill_list<T, FULL, PBC>), dim3(2 * h->num_cus), dim3(SWEEP_WAVES * WAVE), 0, s, a);
}

template <typename T, bool FULL, bool PBC> void launch_sweep_kind(nl_handle_t h, int mode, hipStream_t s) {
  const int32_t owned_layers = h->b_slab ? h->b_mzl - 2 : h->b_mzl;
  const int32_t ncells_i = h->m[0] * h->m[1] * owned_layers;
  if constexpr (sizeof(T) == 4 && !PBC) {
    if (h->b_rows) {
      if (h->b_rows_v == 0) launch_rows<0, FULL>(h, mode, ncells_i, s);
      else if (h->b_rows_v == 1) launch_rows<1, FULL>(h, mode, ncells_i, s);
      else launch_rows<2, FULL>(h, mode, ncells_i, s);
      return;
    }
  }
  const SweepArgs<T> a = sweep_args<T>(h);
  if (h->b_use_masks) {
    if (mode == MODE_COUNT) {
      if constexpr (sizeof(T) == 4) {
        if (!PBC && h->b_mask_nb == 1) {
          // a workgroup per cell, single-batch cells only; the others go on the hand-over list of the batched search
          if (h->b_lean_small) hipLaunchKernelGGL((k_sweep_lean_f32<FULL, 2, LEAN_SMALL_CAP>), dim3(ncells_i), dim3(2 * WAVE), 0, s, a);
          else hipLaunchKernelGGL((k_sweep_lean_f32<FULL>), dim3(ncells_i), dim3(SWEEP_WAVES * WAVE), 0, s, a);
          hipLaunchKernelGGL((k_sweep_list_f32<FULL>), dim3(2 * h->num_cus), dim3(SWEEP_WAVES * WAVE), 0, s, a);
        } else {
          hipLaunchKernelGGL((k_sweep_count_masks_f32<FULL, PBC>), dim3(ncells_i), dim3(SWEEP_WAVES * WAVE), h->dbg_lds_pad, s, a);
        }
      } else
        hipLaunchKernelGGL((k_sweep<T, MODE_COUNT_MASKS, FULL, PBC>), dim3(ncells_i), dim3(SWEEP_WAVES * WAVE), 0, s, a);
    } else if (h->b_wide) {
      launch_fill_masks<T, FULL, PBC, int64_t>(h, a, ncells_i, s);
    } else {
      launch_fill_masks<T, FULL, PBC, int32_t>(h, a, ncells_i, s);
    }
    return;
  }
  // two distance sweeps
  SweepArgs<T> a2 = a;
  a2.isplit = h->b_isplit;
  const int32_t grid2 = ncells_i * h->b_isplit;
  if (mode == MODE_COUNT) {
    if constexpr (sizeof(T) == 4)
      hipLaunchKernelGGL((k_sweep_count_f32<FULL, PBC>), dim3(grid2), dim3(SWEEP_WAVES * WAVE), 0, s, a2);
    else
      hipLaunchKernelGGL((k_sweep<T, MODE_COUNT, FULL, PBC>), dim3(grid2), dim3(SWEEP_WAVES * WAVE), 0, s, a2);
  } else
    hipLaunchKernelGGL((k_sweep<T, MODE_FILL, FULL, PBC>), dim3(grid2), dim3(SWEEP_WAVES * WAVE), 0, s, a2);
}

template <typename T> void launch_sweep(nl_handle_t h, int mode, hipStream_t s) {
  if (h->b_full) {
    if (h->pbc) launch_sweep_kind<T, true, true>(h, mode, s);
    else launch_sweep_kind<T, true, false>(h, mode, s);
  } else {
    if (h->pbc) launch_sweep_kind<T, false, true>(h, mode, s);
    else launch_sweep_kind<T, false, false>(h, mode, s);
  }
}

// Enqueues one whole build. ev != nullptr: records an event before every stage and one after the last.
// part: PART_ALL = the whole build; PART_BEGIN = everything that needs the OWNED particles only (slab builds: memset +
// the binning pass over [0, n_rows)); PART_FINISH = the rest (the binning pass over the ghosts, search, scan,
// expansion).  BEGIN + FINISH = ALL for the caller; between the two the halo exchange may still be writing the ghosts.
enum { PART_ALL = 0, PART_BEGIN = 1, PART_FINISH = 2 };

// Mask rows for `nb` LDS batches per particle: allocated on first need (a half-shell handle that meets a minimum-image
// or dense build; a first dense build).
bool mask_rows_ready(nl_handle_t h, int64_t nb, size_t row_bytes = MASK_ROW_BYTES) {
  const size_t need = row_bytes * (size_t)nb * ((size_t)h->n_max + 64);  // (+64: the expansion kernels read whole row batches)
  if (need > h->dense_masks_limit) return false;
  if (need > h->masks_bytes || !h->masks) {
    if (dev_alloc(h, &h->masks, need) != NL_OK) {
      h->masks_bytes = 0;
      return false;
    }
    h->masks_bytes = need;
  }
  return true;
}

// The fine-row layout needs the two-level binning (k_bin_cells<FINE>) and a fine table that an int32 can index.
bool rows_layout_ok(nl_handle_t h, int32_t mzl) {
  const int64_t nrows = (int64_t)h->m[1] * mzl;
  return h->bin_two_level && nrows <= BIN_MAX_ROWS && h->m[0] <= BIN_FINE_MAX_MX && 4 * (int64_t)h->m[0] * nrows < 2147483000LL;
}
// Two particles five or more quarter-planes apart along z have rounded products t = z * ims more than 1 apart, so their
// distance along z exceeds ms (1 - 8 m 2^-24) (two roundings of t, relative 2^-24 each, at |t| <= 2 m, and the
// rounding of ims): the pair fails the cut-off test in any rounding of r2 once ms / rc > 1 + 8 m 2^-24 + 2^-20.
bool rows_margin_ok(nl_handle_t h) {
  const double ms = h->L[2] / h->m[2];
  return ms / h->rc >= 1.0 + 8.0 * h->m[2] * 5.9604644775390625e-8 + 9.5367431640625e-7;
}

// What the handle remembers about the build being enqueued (also set when a captured graph of it is replayed): how the
// later stages, the getters and a refill after growth have to read the buffers.
template <typename T>
void set_build_state(nl_handle_t h, const void* q_dev, int32_t stride, const int32_t* gid, int32_t n, int32_t z_lo,
                     int32_t mzl, int32_t slab) {
  const int64_t ncl = (int64_t)h->m[0] * h->m[1] * mzl;
  h->ncell_local = ncl;
  h->b_mzl = mzl, h->b_slab = slab, h->b_zlo = z_lo, h->b_stride = stride, h->b_q = q_dev, h->b_gid = gid;
  h->b_full = h->list_kind == NL_LIST_FULL;
  h->b_pbc = h->pbc;
  h->b_variant = h->sweep_variant;
  // Hit masks pay off while a cell's stencil fits one LDS batch; where the mean stencil (27 cells) is close to or
  // beyond the batch size most cells would fall back to a re-search in small batches, so use two full sweeps there.
  const double mean_stream = ncl > 0 ? 27.0 * n / (double)ncl : 0.0;
  const bool sparse_enough = mean_stream <= 0.85 * SweepCfg<T>::CAP;  // mean stencil <= 1088: <= 40.3 per cell
  h->b_use_masks = h->b_variant >= 3 && sparse_enough && mask_rows_ready(h, 1);
  h->b_mask_nb = 1;
  // sparse boxes (mean stream + 5 sigma within half the LDS buffer: up to 19.6 particles per cell): the 2-wave instance
  // of the lean COUNT sweep; a cell beyond it goes to the batched search like any other that does not fit
  h->b_lean_small = h->lean_small_env != 0 && mean_stream + 5.0 * std::sqrt(mean_stream) <= (double)LEAN_SMALL_CAP;
  // The fine-row search (nl_rows.hpp): fp32, open box, the two-level binning, and a cell edge that exceeds the cut-off
  // along z by more than the rounding of the cell hash can hide (rows_margin_ok).  RowsCfg by the mean stencil
  // stream m = 27 <N/cell>: m + 5 sigma within the LDS buffer, the piece a wave walks + 6 sigma within its hit word.
  h->b_rows = false;
  if (sizeof(T) == 4 && h->b_variant >= 3 && !h->pbc && h->rows_env != 0 && rows_layout_ok(h, mzl) && rows_margin_ok(h)) {
    int v = -1;
    if (h->rows_env > 0 && h->rows_env <= 3) {
      v = h->rows_env - 1;
    } else if (h->rows_env == 4 || !sparse_enough) {
      // (at the BASELINE densities the 27-cell sweep is the faster one: 0.512 against 0.570 ms at config 2; from 40.3
      // particles per cell on its streams no longer fit one LDS batch: 0.97 against 0.60 ms at rho = 1.1 --
      // profiles/r03_density_sweep.txt)
      // (a wave walks 27 of the 36 windows)
      const double span = mean_stream * 27.0 / 36.0;
      const int cap[3] = {RowsCfg<0>::CAP, RowsCfg<1>::CAP, RowsCfg<2>::CAP}, bits[3] = {16, 32, 32};
      for (int k = 0; k < 3 && v < 0; k++)
        if (mean_stream + 5.0 * std::sqrt(mean_stream) <= cap[k] && span + 6.0 * std::sqrt(span) <= 64.0 * bits[k]) v = k;
    }
    if (v >= 0 && mask_rows_ready(h, 1, v == 0 ? 128 : 256)) h->b_rows = true, h->b_rows_v = v, h->b_use_masks = true;
  }
  if (h->b_variant >= 3 && !sparse_enough && !h->dense_masks_off) {
    // Dense cells: hit masks for up to FD_NB LDS batches per slot instead of a second distance sweep, when the streams
    // (mean + 5 sigma of a Poisson count) fit that many batches and the mask rows fit the memory set aside for them
    const int64_t nb = (int64_t)((mean_stream + 5.0 * std::sqrt(mean_stream) + 64.0) / SweepCfg<T>::CAP) + 1;
    if (nb <= FD_NB && mask_rows_ready(h, nb)) h->b_use_masks = true, h->b_mask_nb = (int32_t)nb;  // (allocates once)
  }
  // 64-bit list offsets as soon as the list this handle can hold exceeds what an int32 key_pointer can address
  // (the reference's own limit, neighlist_cpu.hpp:15,29); nl_set_offset_width overrides.
  {  // NL_ISPLIT (diagnostics): workgroups per cell in the two-sweep path; default 1 (see sweep_cell)
    int32_t sp = std::max(1, std::min(h->isplit_env, 32));
    if ((int64_t)sp * ncl > 2000000000LL) sp = 1;
    h->b_isplit = h->b_use_masks ? 1 : sp;
  }
  h->b_wide = h->offset_width == 64 || (h->offset_width == 0 && h->capacity > 2147483647LL);
  h->kp_alt_valid = false;
}

template <typename T>
int enqueue_build(nl_handle_t h, const void* q_dev, int32_t stride, const int32_t* gid, int32_t n_rows, int32_t n,
                  int32_t z_lo, int32_t mzl, int32_t slab, hipStream_t s, hipEvent_t* ev, int part = PART_ALL,
                  int32_t n_ghost_lo = 0) {
  const Grid<T> g = make_grid<T>(h, n_rows, z_lo, mzl, slab);
  const int64_t ncl = (int64_t)h->m[0] * h->m[1] * mzl;
  set_build_state<T>(h, q_dev, stride, gid, h->b_dyn ? h->b_n_est : n, z_lo, mzl, slab);
  const int32_t nbp = (n + 255) / 256;
  const T* q = static_cast<const T*>(q_dev);

  const int32_t nrows = h->m[1] * mzl;
  const bool two_level = h->bin_two_level && nrows <= BIN_MAX_ROWS && h->m[0] <= BIN_MAX_MX;
  // One allocation = [cell histogram | status, tickets, total (32 words) | row totals]: one memset node clears
  // what this build's path needs (histogram + meta, or meta + row totals).
  // two binning passes: owned, then ghosts (the persistent sweep, variant 2, has its own cell walk without the guard
  // against an inconsistent cell table: no split there)
  const bool split = part != PART_ALL && two_level && slab;
  if (part != PART_ALL && !split) {  // nothing to overlap on this path: BEGIN does nothing, FINISH is the whole build
    if (part == PART_BEGIN) return NL_OK;
    part = PART_ALL;
  }
  if (two_level) {
    const int32_t my = h->m[1];
    // the three launches of one pass; rc_arr / rs_arr: the pass's own row totals and row starts
    auto run_pass = [&](const BinPhase& ph, int32_t* rc_arr, int32_t* rs_arr, int32_t cells_grid, bool events) {
      const int32_t np = ph.i_end - ph.i_beg;
      const int32_t blocks = std::max(1, (np + h->bin_chunk - 1) / h->bin_chunk);  // (an empty pass still publishes its row starts)
      hipLaunchKernelGGL((k_bin_rows<T>), dim3(blocks), dim3(BIN_THREADS), 0, s, q, stride, n, h->bin_chunk, g, nrows, rc_arr,
                         h->blk_base, h->status, ph);
      if (events) (void)hipEventRecord(ev[NL_STAGE_CELL_SCAN], s);
      // (no scan launch: every block of k_bin_scatter scans the row totals itself and block 0 publishes the row starts)
      if (events) (void)hipEventRecord(ev[NL_STAGE_REORDER], s);
      if (h->bin_chunk >= 8 * BIN_THREADS)
        hipLaunchKernelGGL((k_bin_scatter<T, 8>), dim3(blocks), dim3(BIN_THREADS), 0, s, q, stride, gid, n, h->bin_chunk, g, nrows,
                           rc_arr, rs_arr, h->blk_base, static_cast<Pos<T>*>(h->tmp_pos), h->tmp_row, h->status, ph);
      else
        hipLaunchKernelGGL((k_bin_scatter<T, 4>), dim3(blocks), dim3(BIN_THREADS), 0, s, q, stride, gid, n, h->bin_chunk, g, nrows,
                           rc_arr, rs_arr, h->blk_base, static_cast<Pos<T>*>(h->tmp_pos), h->tmp_row, h->status, ph);
      if constexpr (sizeof(T) == 4) {
        if (h->b_rows) {
          hipLaunchKernelGGL((k_bin_cells<T, true>), dim3(cells_grid), dim3(256), 0, s, g, nrows, rs_arr,
                             static_cast<const Pos<T>*>(h->tmp_pos), h->tmp_row, h->cell_start, static_cast<Pos<T>*>(h->sorted),
                             h->sorted_row, h->sorted_gid, ph);
          return;
        }
      }
      hipLaunchKernelGGL((k_bin_cells<T, false>), dim3(cells_grid), dim3(256), 0, s, g, nrows, rs_arr,
                         static_cast<const Pos<T>*>(h->tmp_pos), h->tmp_row, h->cell_start, static_cast<Pos<T>*>(h->sorted),
                         h->sorted_row, h->sorted_gid, ph);
    };
    if (!split) {
      HIPCHK(h, hipMemsetAsync(h->cell_count + h->ncell, 0, sizeof(int32_t) * (size_t)(32 + nrows), s));
      if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_HASH], s));
      const BinPhase all = {0, n, nrows, 0, 0, 0, nrows, nrows, -1, h->b_dyn};
      run_pass(all, h->row_count, h->row_start, nrows, ev != nullptr);
    } else if (part == PART_BEGIN) {
      // owned particles: rows of the layers 1 .. mzl-2, placed behind the n_ghost_lo particles of ghost layer 0
      HIPCHK(h, hipMemsetAsync(h->cell_count + h->ncell, 0, sizeof(int32_t) * (size_t)(32 + 2 * (size_t)nrows), s));
      const BinPhase owned = {0, n_rows, nrows, n_ghost_lo, n_ghost_lo, my, nrows - 2 * my, nrows, -1};
      run_pass(owned, h->row_count, h->row_start, nrows - 2 * my, false);
      HIPCHK(h, hipGetLastError());
      return NL_OK;
    } else {
      // ghosts: layer 0 at the front of the sorted array, layer mzl-1 behind the owned particles
      const BinPhase ghosts = {n_rows, n, nrows - my, 0, n_rows, 0, my, nrows - my, n_ghost_lo};
      run_pass(ghosts, h->row_count + nrows, h->row_start + nrows + 16, 2 * my, false);
    }
  } else {
    HIPCHK(h, hipMemsetAsync(h->cell_count, 0, sizeof(int32_t) * (size_t)(h->ncell + 32), s));
    if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_HASH], s));
    if (n > 0) hipLaunchKernelGGL((k_hash<T>), dim3(nbp), dim3(256), 0, s, q, stride, n, g, h->cell_count, h->rank, h->status);
    if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_CELL_SCAN], s));
    if (int rc = launch_scan(h, h->cell_count, ncl, h->cell_start, h->totals, s)) return rc;
    if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_REORDER], s));
    if (n > 0)
      hipLaunchKernelGGL((k_reorder<T>), dim3(nbp), dim3(256), 0, s, q, stride, gid, n, g, h->cell_start, h->rank,
                         static_cast<Pos<T>*>(h->sorted), h->sorted_row, h->sorted_gid);
  }
  if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_COUNT], s));
  // (rows of particles rejected by the hash keep a stale count: such a build fails with its status anyway)
  launch_sweep<T>(h, MODE_COUNT, s);
  if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_ROW_SCAN], s));
  if (h->b_wide) {
    if (int rc = launch_scan(h, h->count, n_rows, static_cast<int64_t*>(h->key_pointer), h->totals + 1, s, h->status + META_TOTAL)) return rc;
  } else {
    if (int rc = launch_scan(h, h->count, n_rows, static_cast<int32_t*>(h->key_pointer), h->totals + 1, s, h->status + META_TOTAL)) return rc;
  }
  if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_FILL], s));
  launch_sweep<T>(h, MODE_FILL, s);
  if (ev) HIPCHK(h, hipEventRecord(ev[NL_STAGE_TOTAL], s));
  HIPCHK(h, hipGetLastError());
  return NL_OK;
}

int enqueue_result_copy(nl_handle_t h, hipStream_t s) {
  HIPCHK(h, hipMemcpyAsync(h->host, h->status, sizeof(uint32_t) * META_WORDS, hipMemcpyDeviceToHost, s));
  return NL_OK;
}

int dispatch_build(nl_handle_t h, const void* q, int32_t stride, const int32_t* gid, int32_t n_rows, int32_t n,
                   int32_t z_lo, int32_t mzl, int32_t slab, hipStream_t s, hipEvent_t* ev, int part = PART_ALL,
                   int32_t n_ghost_lo = 0) {
  return h->dtype == NL_F32 ? enqueue_build<float>(h, q, stride, gid, n_rows, n, z_lo, mzl, slab, s, ev, part, n_ghost_lo)
                            : enqueue_build<double>(h, q, stride, gid, n_rows, n, z_lo, mzl, slab, s, ev, part, n_ghost_lo);
}

// Default list capacity (unless the caller fixed it): ideal-gas estimate of the half-pair count
// N * rho * (2/3) pi rc^3 with 30 % head room, twice that for a full list.
int estimate_capacity(nl_handle_t h) {
  if (h->capacity_user) return NL_OK;
  const double rho = (double)h->n_max / (h->L[0] * h->L[1] * h->L[2]);
  const double per = rho * (2.0 / 3.0) * 3.14159265358979323846 * h->rc * h->rc * h->rc;
  int64_t want = (int64_t)((double)h->n_max * per * 1.3) + 64 * (int64_t)h->n_max + 4096;
  if (h->list_kind == NL_LIST_FULL) want *= 2;
  if (want > h->capacity) {
    if (int rc = dev_alloc(h, &h->list, 4 * (size_t)want)) return rc;
    h->capacity = want;
  }
  return NL_OK;
}

int grow_list(nl_handle_t h, int64_t need) {
  int64_t cap = std::max<int64_t>(need + need / 8 + 1024, h->capacity);
  // a list that an int32 key_pointer can still address stays below the switch to 64-bit offsets
  if (need <= 2147483647LL && cap > 2147483647LL && h->capacity <= 2147483647LL) cap = 2147483647LL;
  int rc = dev_alloc(h, &h->list, sizeof(int32_t) * (size_t)cap);
  if (rc) {
    h->capacity = 0;
    return rc;
  }
  h->capacity = cap;
  return NL_OK;
}

// Waits for the pending build; in a synchronous build an undersized list is grown and the fill pass re-run.
int finish(nl_handle_t h, bool may_grow) {
  if (!h->pending) return h->built ? NL_OK : fail(h, NL_ERR_STATE);
  HIPCHK(h, hipStreamSynchronize(h->last_stream));
  h->pending = false;
  uint32_t st = h->host->status;
  if ((st & ST_INDEX_OVERFLOW) && !(st & ~(ST_CAPACITY | ST_INDEX_OVERFLOW)) && may_grow && !h->b_wide && h->offset_width == 0) {
    // more than INT32_MAX entries in a build with 32-bit offsets: the list grows past that size, which makes builds of
    // this handle wide (64-bit key_pointer), and the whole build runs again
    int rc = grow_list(h, h->host->total());
    if (rc) return rc;
    rc = dispatch_build(h, h->b_q, h->b_stride, h->b_gid, h->n_rows, h->n, h->b_zlo, h->b_mzl, h->b_slab, h->last_stream, nullptr);
    if (!rc) rc = enqueue_result_copy(h, h->last_stream);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->last_stream));
    st = h->host->status;
  } else if ((st & ST_CAPACITY) && !(st & ~ST_CAPACITY) && may_grow) {
    int rc = grow_list(h, h->host->total());
    if (rc) return rc;
    HIPCHK(h, hipMemsetAsync(h->status, 0, sizeof(uint32_t), h->last_stream));
    if (h->dtype == NL_F32)
      launch_sweep<float>(h, MODE_FILL, h->last_stream);
    else
      launch_sweep<double>(h, MODE_FILL, h->last_stream);
    rc = enqueue_result_copy(h, h->last_stream);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->last_stream));
    st = h->host->status;
  }
  int err = status_to_error(st);
  if (h->b_dyn && h->dyn_host) {  // a decomposed build: its ghost counts, and whether the exchange held what was sent
    h->n = h->n_rows + h->dyn_host[2] + h->dyn_host[3];
    if (!err && h->dyn_host[4]) err = NL_ERR_CAPACITY;
  }
  h->built = err == NL_OK;
  if (err) return fail(h, err);
  return NL_OK;
}

}  // namespace

namespace {
template <typename SRC, typename DST>
__global__ void __launch_bounds__(256) k_convert_offsets(const SRC* __restrict__ in, DST* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (DST)in[i];
}

// key_pointer of the last build in the requested width (32 or 64): the buffer the build wrote when the widths agree,
// else a converted copy made once per build.  A wide list that an int32 cannot address is NL_ERR_INDEX_OVERFLOW.
int key_pointer_as(nl_handle_t h, int width, const void** out) {
  const bool want_wide = width == 64;
  if (want_wide == h->b_wide) {
    *out = h->key_pointer;
    return NL_OK;
  }
  if (!want_wide && h->host->total() > 2147483647LL) return fail(h, NL_ERR_INDEX_OVERFLOW);
  const int64_t cnt = (int64_t)h->n_rows + 1;
  if (!h->kp_alt) {
    HIPCHK(h, hipSetDevice(h->device));
    void* p = nullptr;
    if (hipMalloc(&p, 8 * ((size_t)h->n_max + 32)) != hipSuccess) return fail(h, NL_ERR_NOMEM);
    h->kp_alt = p;
    h->kp_alt_valid = false;
  }
  if (!h->kp_alt_valid) {
    const int32_t nb = (int32_t)((cnt + 255) / 256);
    if (want_wide)
      hipLaunchKernelGGL((k_convert_offsets<int32_t, int64_t>), dim3(nb), dim3(256), 0, h->last_stream,
                         static_cast<const int32_t*>(h->key_pointer), static_cast<int64_t*>(h->kp_alt), cnt);
    else
      hipLaunchKernelGGL((k_convert_offsets<int64_t, int32_t>), dim3(nb), dim3(256), 0, h->last_stream,
                         static_cast<const int64_t*>(h->key_pointer), static_cast<int32_t*>(h->kp_alt), cnt);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->last_stream));
    h->kp_alt_valid = true;
  }
  *out = h->kp_alt;
  return NL_OK;
}

// Order-independent checksum of the list: sum over entries (row i, partner j) of mix((id_i << 32) | j), mix(v): v *= 0x9E3779B97F4A7C15,
// v ^= v >> 29 (wrapping) -- the pair-set hash the known answers of SURVEY.md section 8c are stored as.  One wave per row at a time.
template <typename T, typename OFF>
__global__ void __launch_bounds__(256) k_list_checksum(const OFF* __restrict__ kp, const int32_t* __restrict__ list, int32_t n_rows,
                                                       const int32_t* __restrict__ gid, const T* __restrict__ q,
                                                       unsigned long long* __restrict__ acc) {
  __shared__ unsigned long long part[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long h = 0;
  for (int64_t row = (int64_t)blockIdx.x * 4 + w; row < n_rows; row += (int64_t)gridDim.x * 4) {
    uint32_t id = (uint32_t)row;
    if (gid == reinterpret_cast<const int32_t*>(1)) {  // NL_GID_IN_W
      if constexpr (sizeof(T) == 4) id = (uint32_t)__float_as_int(q[(size_t)row * 4 + 3]);
      else id = (uint32_t)__double_as_longlong(q[(size_t)row * 4 + 3]);
    } else if (gid) {
      id = (uint32_t)gid[row];
    }
    const OFF b = kp[row], e = kp[row + 1];
    for (OFF k = b + lane; k < e; k += 64) {
      unsigned long long v = ((unsigned long long)id << 32) | (uint32_t)list[k];
      v *= 0x9E3779B97F4A7C15ULL;
      v ^= v >> 29;
      h += v;
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) h += __shfl_xor(h, d, 64);
  if (lane == 0) part[w] = h;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

// CopyGather (neighlist_gpu.hpp:144-151) / Gather + SortPtclData (neighlist_cpu.hpp:170-180): dst[s] = src[order[s]]
// for elements of W 32-bit words.
template <int W>
__global__ void __launch_bounds__(256) k_gather_words(const uint32_t* __restrict__ src, const int32_t* __restrict__ order, int32_t n,
                                                      uint32_t* __restrict__ dst) {
  const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t* p = src + (size_t)order[s] * W;
  uint32_t* d = dst + (size_t)s * W;
  if constexpr (W % 4 == 0) {
#pragma unroll
    for (int k = 0; k < W / 4; k++) reinterpret_cast<uint4*>(d)[k] = reinterpret_cast<const uint4*>(p)[k];
  } else if constexpr (W % 2 == 0) {
#pragma unroll
    for (int k = 0; k < W / 2; k++) reinterpret_cast<uint2*>(d)[k] = reinterpret_cast<const uint2*>(p)[k];
  } else {
#pragma unroll
    for (int k = 0; k < W; k++) d[k] = p[k];
  }
}

int get_csr(nl_handle_t h, bool full, int width, const void** key_pointer_dev, const int32_t** list_dev,
            const int32_t** number_of_partners_dev, int64_t* nentries) {
  if (!h) return NL_ERR_ARG;
  int rc = nl_synchronize(h);
  if (rc) return rc;
  if (h->b_full != full) return fail(h, NL_ERR_STATE);
  if (key_pointer_dev)
    if ((rc = key_pointer_as(h, width, key_pointer_dev))) return rc;
  if (!key_pointer_dev && width == 32 && h->host->total() > 2147483647LL) return fail(h, NL_ERR_INDEX_OVERFLOW);
  if (list_dev) *list_dev = h->list;
  if (number_of_partners_dev) *number_of_partners_dev = h->count;
  if (nentries) *nentries = h->host->total();
  return NL_OK;
}
}  // namespace


extern "C" {

const char* nl_status_string(int s) {
  switch (s) {
    case NL_OK: return "ok";
    case NL_ERR_ARG: return "bad argument";
    case NL_ERR_NOMEM: return "out of memory";
    case NL_ERR_OUT_OF_BOX: return "particle more than one box length outside the box (or NaN)";
    case NL_ERR_CAPACITY: return "pair list capacity exceeded";
    case NL_ERR_HIP: return "HIP runtime error";
    case NL_ERR_STATE: return "call order violated";
    case NL_ERR_MESH: return "fewer than 3 cells along an axis";
    case NL_ERR_INDEX_OVERFLOW: return "more than INT32_MAX list entries behind a 32-bit key_pointer";
    case NL_ERR_NO_DEVICE: return "no usable HIP device";
    case NL_ERR_DOMAIN: return "particle outside the layers declared for this rank";
    case NL_ERR_COMM: return "communication failed (RCCL not loadable, communicator or transport error)";
    default: return "unknown status";
  }
}

int nl_device_count(int* count) {
  if (!count) return NL_ERR_ARG;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *count = c;
  return NL_OK;
}

int nl_create(nl_handle_t* out, int dtype, double rc, double Lx, double Ly, double Lz, int device_id) {
  if (!out) return NL_ERR_ARG;
  *out = nullptr;
  if ((dtype != NL_F32 && dtype != NL_F64) || !(rc > 0) || !(Lx > 0) || !(Ly > 0) || !(Lz > 0)) return NL_ERR_ARG;
  const double L[3] = {Lx, Ly, Lz};
  int32_t m[3];
  for (int d = 0; d < 3; d++) {
    const double r = L[d] / rc;
    if (!(r < 2147483647.0)) return NL_ERR_ARG;
    m[d] = (int32_t)r;  // neighlist_cpu.hpp:384-386
    if (m[d] < 3) return NL_ERR_MESH;
  }
  if ((double)m[0] * m[1] * m[2] > 2.0e9) return NL_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NL_ERR_NO_DEVICE;
  int dev = device_id;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return NL_ERR_NO_DEVICE;
  if (dev >= ndev) return NL_ERR_ARG;
  if (hipSetDevice(dev) != hipSuccess) return NL_ERR_NO_DEVICE;

  nl_handle_t h = new (std::nothrow) nl_handle_s();
  if (!h) return NL_ERR_NOMEM;
  h->dtype = dtype, h->device = dev, h->rc = rc;
  h->rc2 = rc * rc;  // neighlist_cpu.hpp:394 (double)
  h->rc2_f = floor_to_float(h->rc2);
  for (int d = 0; d < 3; d++) {
    h->L[d] = L[d], h->m[d] = m[d];
    // ms_ and ims_ live in a Vec of the position type (neighlist_cpu.hpp:12,389-391,409-411)
    const float ms_f = (float)(L[d] / m[d]);
    h->ms_f[d] = ms_f;
    h->ims_f[d] = (float)(1.0 / (double)ms_f);
    const double ms_d = L[d] / m[d];
    h->ims_d[d] = 1.0 / ms_d;
  }
  h->ncell = (int64_t)m[0] * m[1] * m[2];
  if (hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&h->host), sizeof(HostResult), hipHostMallocDefault) != hipSuccess) {
    nl_destroy(h);
    return NL_ERR_HIP;
  }
  for (auto& e : h->ev)
    if (hipEventCreate(&e) != hipSuccess) {
      nl_destroy(h);
      return NL_ERR_HIP;
    }
  memset(h->host, 0, sizeof(HostResult));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) h->num_cus = prop.multiProcessorCount;
    if (const char* v = getenv("NL_SWEEP_VARIANT")) h->sweep_variant = atoi(v) <= 1 ? 1 : 3;
    if (const char* v = getenv("NL_ISPLIT")) h->isplit_env = std::max(0, atoi(v));
    if (const char* v = getenv("NL_DENSE_MASKS")) h->dense_masks_off = atoi(v) == 0;
    if (const char* v = getenv("NL_ROWS")) h->rows_env = std::max(-1, std::min(atoi(v), 4));
    if (const char* v = getenv("NL_LEAN_SMALL")) h->lean_small_env = atoi(v) != 0;
    if (const char* v = getenv("NL_FILL_SMALL")) h->fill_small_env = atoi(v) != 0;
    if (const char* v = getenv("NL_OFFSET_WIDTH")) h->offset_width = atoi(v) == 64 ? 64 : atoi(v) == 32 ? 32 : 0;
    if (const char* v = getenv("NL_BINNING")) h->bin_two_level = atoi(v) != 1;
    if (const char* v = getenv("NL_GRAPH")) h->use_graph = atoi(v) != 0;
    if (const char* v = getenv("NL_DEBUG_FLAGS")) h->dbg_flags = atoi(v);
    if (const char* v = getenv("NL_DEBUG_WG_PER_CU")) h->dbg_wg_per_cu = std::max(1, atoi(v));
    if (const char* v = getenv("NL_DEBUG_LDS_PAD")) h->dbg_lds_pad = std::max(0, atoi(v));
  }
  *out = h;
  return NL_OK;
}

int nl_destroy(nl_handle_t h) {
  if (!h) return NL_ERR_ARG;
  (void)hipSetDevice(h->device);
  if (h->pending && h->last_stream) (void)hipStreamSynchronize(h->last_stream);
  void* bufs[] = {h->rank, h->sorted, h->sorted_row, h->sorted_gid, h->count, h->key_pointer, h->kp_alt, h->progress, h->base_sorted, h->row_start, h->blk_base, h->tmp_pos, h->tmp_row, h->masks, h->full27_list, h->resort_buf, h->dbg_buf, h->cell_count,
                  h->cell_start, h->scan_look, h->totals, h->list, h->t_list, h->t_count, h->t_cursor};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (h->host) (void)hipHostFree(h->host);
  for (auto& e : h->ev)
    if (e) (void)hipEventDestroy(e);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return NL_OK;
}

int nl_initialize(nl_handle_t h, int32_t n_max) {
  if (!h || n_max < 0) return fail(h, NL_ERR_ARG);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) {
    int rc = finish(h, false);
    (void)rc;
  }
  h->built = false;
  const size_t n = (size_t)n_max;
  const size_t pos_bytes = h->dtype == NL_F32 ? sizeof(Pos<float>) : sizeof(Pos<double>);
  int rc;
  if ((rc = dev_alloc(h, &h->rank, 4 * (n + 16)))) return rc;
  if ((rc = dev_alloc(h, &h->sorted, pos_bytes * (n + 16)))) return rc;
  if ((rc = dev_alloc(h, &h->sorted_row, 4 * (n + 64)))) return rc;  // (+64: k_fill_masks reads whole row batches)
  if ((rc = dev_alloc(h, &h->sorted_gid, 4 * (n + 16)))) return rc;
  if ((rc = dev_alloc(h, &h->count, 4 * (n + 32)))) return rc;
  if ((rc = dev_alloc(h, &h->key_pointer, 8 * (n + 32)))) return rc;  // int32 or int64 offsets (b_wide)
  if (h->kp_alt) (void)hipFree(h->kp_alt), h->kp_alt = nullptr;
  h->kp_alt_valid = false;
  if (h->resort_buf) (void)hipFree(h->resort_buf), h->resort_buf = nullptr;
  if ((rc = dev_alloc(h, &h->progress, 4 * (n + 16)))) return rc;
  if ((rc = dev_alloc(h, &h->base_sorted, 8 * (n + 64)))) return rc;  // (dense builds only: k_fill_dense)
  {
    const size_t nrows = (size_t)h->m[1] * h->m[2];
    // chunk per block: 4096 particles, 8192 from half a million on (cfg 2: binning 60.7 -> 55.9 us, cfg 3 64.7 -> 58.0;
    // 16384: 65.7), more for very large N so that blk_base stays small
    h->bin_chunk = n >= (1 << 19) ? 8192 : 4096;
    if (const char* v = getenv("NL_DEBUG_BIN_CHUNK")) h->bin_chunk = std::max(1024, atoi(v));  // diagnostics
    while ((n + h->bin_chunk - 1) / h->bin_chunk > 1024) h->bin_chunk *= 2;
    h->bin_blocks = (int32_t)((n + h->bin_chunk - 1) / h->bin_chunk);
    if (h->bin_blocks < 1) h->bin_blocks = 1;
    if ((rc = dev_alloc(h, &h->row_start, 4 * (2 * nrows + 64)))) return rc;  // (two arrays: a split slab build has two passes)
    if ((rc = dev_alloc(h, &h->blk_base, 4 * (nrows * (size_t)h->bin_blocks + 16)))) return rc;
    if ((rc = dev_alloc(h, &h->tmp_pos, pos_bytes * (n + 16)))) return rc;
    if ((rc = dev_alloc(h, &h->tmp_row, 4 * (n + 16)))) return rc;
  }
  if (h->sweep_variant >= 3) {
    if ((rc = dev_alloc(h, &h->masks, (size_t)MASK_ROW_BYTES * (n + 64)))) return rc;
    h->masks_bytes = (size_t)MASK_ROW_BYTES * (n + 64);
  }
  // cells handed from one search kernel to another: half-shell -> 27-cell search, pipelined COUNT -> batched search
  if ((rc = dev_alloc(h, &h->full27_list, 4 * ((size_t)h->ncell + 16)))) return rc;
  if ((rc = dev_alloc(h, &h->dbg_buf, 8 * (64 + 4 * 4096)))) return rc;
  HIPCHK(h, hipMemset(h->dbg_buf, 0, 8 * (64 + 4 * 4096)));
  if ((rc = dev_alloc(h, &h->cell_count, 4 * ((size_t)h->ncell + 64 + 2 * (size_t)h->m[1] * h->m[2])))) return rc;
  h->row_count = h->cell_count + h->ncell + 32;
  if ((rc = dev_alloc(h, &h->cell_start, 4 * (4 * (size_t)h->ncell + 32)))) return rc;  // (cell_start, or the fine-row table: 4 M + 1)
  const size_t nblk = std::max<size_t>(n, (size_t)h->ncell) / SCAN_BLOCK + 2;
  if ((rc = dev_alloc(h, &h->scan_look, 8 * (nblk + 1)))) return rc;
  HIPCHK(h, hipMemset(h->scan_look, 0, 8 * (nblk + 1)));
  h->scan_blocks = (int32_t)nblk;
  if ((rc = dev_alloc(h, &h->totals, 8 * 4))) return rc;
  h->status = reinterpret_cast<uint32_t*>(h->cell_count + h->ncell);  // cleared by the same memset as the histogram
  HIPCHK(h, hipMemset(h->totals, 0, 32));
  HIPCHK(h, hipMemset(h->cell_count, 0, 4 * ((size_t)h->ncell + 64 + 2 * (size_t)h->m[1] * h->m[2])));
  h->n_max = n_max;
  if ((rc = estimate_capacity(h))) return rc;
  h->t_valid = false;
  return NL_OK;
}

int nl_set_periodic(nl_handle_t h, int minimum_image) {
  if (!h) return NL_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) (void)finish(h, false);
  if ((minimum_image != 0) != h->pbc) {
    h->pbc = minimum_image != 0;
    h->built = false;
    h->t_valid = false;
  }
  return NL_OK;
}

int nl_set_list_kind(nl_handle_t h, int kind) {
  if (!h || (kind != NL_LIST_HALF && kind != NL_LIST_FULL)) return fail(h, NL_ERR_ARG);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) (void)finish(h, false);
  if (kind != h->list_kind) {
    h->list_kind = kind;
    h->built = false;
    h->t_valid = false;
    h->t_rows_cap = 0;  // the transposed buffer is re-allocated (and -1 filled) for the other kind
    if (h->n_max > 0)
      if (int rc = estimate_capacity(h)) return rc;
  }
  return NL_OK;
}

int nl_set_graph(nl_handle_t h, int on) {
  if (!h) return NL_ERR_ARG;
  h->use_graph = on != 0;
  return NL_OK;
}

int nl_set_capacity(nl_handle_t h, int64_t max_pairs) {
  if (!h || max_pairs < 0) return fail(h, NL_ERR_ARG);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) (void)finish(h, false);
  h->built = false;
  int rc = dev_alloc(h, &h->list, 4 * (size_t)(max_pairs + 16));
  if (rc) {
    h->capacity = 0;
    return rc;
  }
  h->capacity = max_pairs;
  h->capacity_user = true;
  return NL_OK;
}

namespace {
// part = PART_ALL: the whole build.  PART_BEGIN: validate, remember the arguments, enqueue what needs only the owned
// particles.  PART_FINISH: enqueue the rest with the remembered arguments.
int make_list_slab_part(nl_handle_t h, const void* q_dev, int32_t q_stride, const int32_t* gid_dev, int32_t n_rows,
                        int32_t n, int32_t n_ghost_lo, int32_t z_lo, int32_t z_hi, void* stream, int sync, int part) {
  if (!h) return NL_ERR_ARG;
  if (part == PART_FINISH) {
    if (!h->begun) return fail(h, NL_ERR_STATE);
    // the second half reuses what the first half is still writing (row totals, the owned region of the sorted array):
    // it must be ordered behind it, i.e. enqueued on the same stream
    if ((hipStream_t)stream != h->last_stream) return fail(h, NL_ERR_STATE);
    q_dev = h->b_q, q_stride = h->b_stride, gid_dev = h->b_gid, n_rows = h->n_rows, n = h->n, n_ghost_lo = h->begun_ghost_lo;
    z_lo = h->begun_zlo, z_hi = h->begun_zhi;
  }
  h->begun = false;
  if (h->n_max <= 0 && n > 0) return fail(h, NL_ERR_STATE);
  if (n < 0 || n_rows < 0 || n_rows > n || n > h->n_max || (q_stride != 3 && q_stride != 4) || (!q_dev && n > 0))
    return fail(h, NL_ERR_ARG);
  if (n_ghost_lo < 0 || n_ghost_lo > n - n_rows) return fail(h, NL_ERR_ARG);
  if (gid_dev == NL_GID_IN_W && q_stride != 4) return fail(h, NL_ERR_ARG);
  const int32_t mz = h->m[2];
  if (z_lo < 0 || z_hi > mz || z_lo >= z_hi) return fail(h, NL_ERR_ARG);
  const int32_t owned = z_hi - z_lo;
  int32_t slab = 1, mzl = owned + 2;
  if (owned == mz) {
    slab = 0, mzl = mz;
    if (n_rows != n) return fail(h, NL_ERR_ARG);
  } else if (mz - owned < 2) {
    return fail(h, NL_ERR_ARG);  // the two ghost layers would be the same layer
  }
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) {
    // back-to-back asynchronous builds (the reference's timing loop): errors of the previous one are dropped,
    // exactly like its results; stream order keeps the buffers consistent when the stream is the same.
    if (h->last_stream != (hipStream_t)stream) HIPCHK(h, hipStreamSynchronize(h->last_stream));
    h->pending = false;
  }
  // The build runs on exactly the stream it is given; NULL is HIP's null (default) stream, as in the reference
  // (make_list.cu:124-127 launches on the default stream), NOT a private stream: work the caller has queued on that
  // stream before the call -- e.g. the kernel that wrote the positions, or a halo exchange -- is finished first.
  hipStream_t s = (hipStream_t)stream;
  h->built = false;
  h->t_valid = false;
  h->n = n, h->n_rows = n_rows;
  int rc;
  if (part == PART_BEGIN) {
    rc = dispatch_build(h, q_dev, q_stride, gid_dev, n_rows, n, z_lo, mzl, slab, s, nullptr, PART_BEGIN, n_ghost_lo);
    if (rc) return rc;
    h->begun = true, h->begun_ghost_lo = n_ghost_lo, h->begun_zlo = z_lo, h->begun_zhi = z_hi;
    h->last_stream = s;
    return NL_OK;
  }
  if (h->use_graph && part == PART_ALL) {
    nl_handle_s::GraphKey key;
    key.q = q_dev, key.gid = gid_dev, key.stride = q_stride, key.n_rows = n_rows, key.n = n, key.z_lo = z_lo, key.mzl = mzl;
    key.slab = slab, key.list_kind = h->list_kind, key.pbc = h->pbc ? 1 : 0, key.capacity = h->capacity;
    key.epoch = h->buffers_epoch, key.offset_width = h->offset_width;
    if (!h->graph_exec || !(key == h->graph_key)) {
      if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec), h->graph_exec = nullptr;
      if (h->graph) (void)hipGraphDestroy(h->graph), h->graph = nullptr;
      // (whatever the build has to allocate -- the mask rows of a first dense build -- is allocated before the capture)
      if (h->dtype == NL_F32) set_build_state<float>(h, q_dev, q_stride, gid_dev, n, z_lo, mzl, slab);
      else set_build_state<double>(h, q_dev, q_stride, gid_dev, n, z_lo, mzl, slab);
      key.epoch = h->buffers_epoch;
      // captured on the private stream (the null stream cannot be captured); replayed on the caller's stream
      HIPCHK(h, hipStreamBeginCapture(h->own_stream, hipStreamCaptureModeRelaxed));
      rc = dispatch_build(h, q_dev, q_stride, gid_dev, n_rows, n, z_lo, mzl, slab, h->own_stream, nullptr);
      if (!rc) rc = enqueue_result_copy(h, h->own_stream);
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(h->own_stream, &g);
      if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
      }
      HIPCHK(h, e);
      h->graph = g;
      HIPCHK(h, hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
      h->graph_key = key;
    }
    else {  // replay: the handle's per-build state is the captured build's, whatever ran in between
      if (h->dtype == NL_F32) set_build_state<float>(h, q_dev, q_stride, gid_dev, n, z_lo, mzl, slab);
      else set_build_state<double>(h, q_dev, q_stride, gid_dev, n, z_lo, mzl, slab);
    }
    HIPCHK(h, hipGraphLaunch(h->graph_exec, s));
  } else {
    rc = dispatch_build(h, q_dev, q_stride, gid_dev, n_rows, n, z_lo, mzl, slab, s, nullptr, part, n_ghost_lo);
    if (rc) return rc;
    rc = enqueue_result_copy(h, s);
    if (rc) return rc;
  }
  h->last_stream = s;
  h->pending = true;
  if (sync) return finish(h, true);
  return NL_OK;
}
}  // namespace

int nl_make_list_slab(nl_handle_t h, const void* q_dev, int32_t q_stride, const int32_t* gid_dev, int32_t n_rows,
                      int32_t n, int32_t z_lo, int32_t z_hi, void* stream, int sync) {
  if (h) h->b_dyn = nullptr, h->dyn_host = nullptr;
  return make_list_slab_part(h, q_dev, q_stride, gid_dev, n_rows, n, 0, z_lo, z_hi, stream, sync, PART_ALL);
}

int nl_make_list_slab_begin(nl_handle_t h, const void* q_dev, int32_t q_stride, const int32_t* gid_dev, int32_t n_rows,
                            int32_t n, int32_t n_ghost_lo, int32_t z_lo, int32_t z_hi, void* stream) {
  if (h) h->b_dyn = nullptr, h->dyn_host = nullptr;
  return make_list_slab_part(h, q_dev, q_stride, gid_dev, n_rows, n, n_ghost_lo, z_lo, z_hi, stream, 0, PART_BEGIN);
}

int nl_make_list_slab_finish(nl_handle_t h, void* stream, int sync) {
  return make_list_slab_part(h, nullptr, 0, nullptr, 0, 0, 0, 0, 0, stream, sync, PART_FINISH);
}

int nl_make_list(nl_handle_t h, const void* q_dev, int32_t q_stride, int32_t n, void* stream, int sync) {
  if (!h) return NL_ERR_ARG;
  return nl_make_list_slab(h, q_dev, q_stride, nullptr, n, n, 0, h->m[2], stream, sync);
}

int nl_synchronize(nl_handle_t h) {
  if (!h) return NL_ERR_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->pending && !h->built) return h->last_error ? h->last_error : fail(h, NL_ERR_STATE);
  return finish(h, false);
}

int nl_get_full_csr(nl_handle_t h, const int32_t** key_pointer_dev, const int32_t** list_dev,
                    const int32_t** number_of_partners_dev, int64_t* nentries) {
  return get_csr(h, true, 32, reinterpret_cast<const void**>(key_pointer_dev), list_dev, number_of_partners_dev, nentries);
}

int nl_get_half_csr(nl_handle_t h, const int32_t** key_pointer_dev, const int32_t** sorted_list_dev,
                    const int32_t** number_of_partners_dev, int64_t* npairs) {
  return get_csr(h, false, 32, reinterpret_cast<const void**>(key_pointer_dev), sorted_list_dev, number_of_partners_dev, npairs);
}

int nl_get_full_csr64(nl_handle_t h, const int64_t** key_pointer_dev, const int32_t** list_dev,
                      const int32_t** number_of_partners_dev, int64_t* nentries) {
  return get_csr(h, true, 64, reinterpret_cast<const void**>(key_pointer_dev), list_dev, number_of_partners_dev, nentries);
}

int nl_get_half_csr64(nl_handle_t h, const int64_t** key_pointer_dev, const int32_t** sorted_list_dev,
                      const int32_t** number_of_partners_dev, int64_t* npairs) {
  return get_csr(h, false, 64, reinterpret_cast<const void**>(key_pointer_dev), sorted_list_dev, number_of_partners_dev, npairs);
}

int nl_list_checksum(nl_handle_t h, uint64_t* checksum, int64_t* nentries) {
  if (!h || !checksum) return fail(h, NL_ERR_ARG);
  int rc = nl_synchronize(h);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t s = h->last_stream;
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(h->totals + 3);
  HIPCHK(h, hipMemsetAsync(acc, 0, 8, s));
  const int32_t n = h->n_rows;
  if (n > 0) {
    const int32_t grid = std::max(1, std::min((n + 3) / 4, 8 * h->num_cus));
    const bool w = h->b_wide, f32 = h->dtype == NL_F32;
    if (f32 && !w)
      hipLaunchKernelGGL((k_list_checksum<float, int32_t>), dim3(grid), dim3(256), 0, s, static_cast<const int32_t*>(h->key_pointer), h->list, n, h->b_gid, static_cast<const float*>(h->b_q), acc);
    else if (f32)
      hipLaunchKernelGGL((k_list_checksum<float, int64_t>), dim3(grid), dim3(256), 0, s, static_cast<const int64_t*>(h->key_pointer), h->list, n, h->b_gid, static_cast<const float*>(h->b_q), acc);
    else if (!w)
      hipLaunchKernelGGL((k_list_checksum<double, int32_t>), dim3(grid), dim3(256), 0, s, static_cast<const int32_t*>(h->key_pointer), h->list, n, h->b_gid, static_cast<const double*>(h->b_q), acc);
    else
      hipLaunchKernelGGL((k_list_checksum<double, int64_t>), dim3(grid), dim3(256), 0, s, static_cast<const int64_t*>(h->key_pointer), h->list, n, h->b_gid, static_cast<const double*>(h->b_q), acc);
    HIPCHK(h, hipGetLastError());
  }
  unsigned long long out = 0;
  HIPCHK(h, hipMemcpyAsync(&out, acc, 8, hipMemcpyDeviceToHost, s));
  HIPCHK(h, hipStreamSynchronize(s));
  *checksum = (uint64_t)out;
  if (nentries) *nentries = h->host->total();
  return NL_OK;
}

int nl_get_cell_order(nl_handle_t h, const int32_t** order_dev, int32_t* n) {
  if (!h) return NL_ERR_ARG;
  int rc = nl_synchronize(h);
  if (rc) return rc;
  if (order_dev) *order_dev = h->sorted_row;
  if (n) *n = h->n;
  return NL_OK;
}

int nl_resort(nl_handle_t h, void* array_dev, size_t elem_bytes, void* stream) {
  if (!h || !array_dev || elem_bytes == 0 || elem_bytes % 4 != 0 || elem_bytes > 32 || elem_bytes == 20 || elem_bytes == 28)
    return fail(h, NL_ERR_ARG);
  int rc = nl_synchronize(h);  // the permutation is the last build's
  if (rc) return rc;
  if (h->b_slab || h->n_rows != h->n) return fail(h, NL_ERR_STATE);  // a permutation of the caller's own particles
  HIPCHK(h, hipSetDevice(h->device));
  const int32_t n = h->n;
  if (n == 0) return NL_OK;
  if (!h->resort_buf) {
    void* p = nullptr;
    if (hipMalloc(&p, 32 * ((size_t)h->n_max + 16)) != hipSuccess) return fail(h, NL_ERR_NOMEM);
    h->resort_buf = p;
  }
  hipStream_t s = (hipStream_t)stream;
  if (s != h->last_stream) HIPCHK(h, hipStreamSynchronize(h->last_stream));
  const uint32_t* src = static_cast<const uint32_t*>(array_dev);
  uint32_t* buf = static_cast<uint32_t*>(h->resort_buf);
  const dim3 grid((n + 255) / 256), block(256);
  switch (elem_bytes / 4) {
    case 1: hipLaunchKernelGGL(k_gather_words<1>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
    case 2: hipLaunchKernelGGL(k_gather_words<2>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
    case 3: hipLaunchKernelGGL(k_gather_words<3>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
    case 4: hipLaunchKernelGGL(k_gather_words<4>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
    case 6: hipLaunchKernelGGL(k_gather_words<6>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
    default: hipLaunchKernelGGL(k_gather_words<8>, grid, block, 0, s, src, h->sorted_row, n, buf); break;
  }
  HIPCHK(h, hipGetLastError());
  HIPCHK(h, hipMemcpyAsync(array_dev, buf, elem_bytes * (size_t)n, hipMemcpyDeviceToDevice, s));
  return NL_OK;
}

int nl_set_offset_width(nl_handle_t h, int bits) {
  if (!h || (bits != 0 && bits != 32 && bits != 64)) return fail(h, NL_ERR_ARG);
  HIPCHK(h, hipSetDevice(h->device));
  if (h->pending) (void)finish(h, false);
  if (bits != h->offset_width) {
    h->offset_width = bits;
    h->built = false;
    h->t_valid = false;
  }
  return NL_OK;
}

int nl_number_of_pairs(nl_handle_t h, int64_t* npairs) {
  if (!h || !npairs) return fail(h, NL_ERR_ARG);
  int rc = nl_synchronize(h);
  if (rc) return rc;
  *npairs = h->b_full ? h->host->total() / 2 : h->host->total();
  return NL_OK;
}

int nl_get_mesh(nl_handle_t h, int32_t mesh[3], int64_t* ncell) {
  if (!h) return NL_ERR_ARG;
  if (mesh)
    for (int d = 0; d < 3; d++) mesh[d] = h->m[d];
  if (ncell) *ncell = h->ncell;
  return NL_OK;
}

int nl_get_sorted(nl_handle_t h, const int32_t** cell_start_dev, const void** sorted_pos_dev,
                  const int32_t** sorted_row_dev, int64_t* ncell_local) {
  if (!h) return NL_ERR_ARG;
  int rc = nl_synchronize(h);
  if (rc) return rc;
  if (cell_start_dev) *cell_start_dev = h->cell_start;
  if (sorted_pos_dev) *sorted_pos_dev = h->sorted;
  if (sorted_row_dev) *sorted_row_dev = h->sorted_row;
  if (ncell_local) *ncell_local = h->ncell_local;
  return NL_OK;
}

int nl_debug_read(nl_handle_t h, uint64_t* out, int32_t n, int reset) {
  if (!h || !out || !h->dbg_buf || n < 0 || n > 64 + 4 * 4096) return NL_ERR_ARG;
  HIPCHK(h, hipDeviceSynchronize());
  HIPCHK(h, hipMemcpy(out, h->dbg_buf, 8 * (size_t)n, hipMemcpyDeviceToHost));
  if (reset) HIPCHK(h, hipMemset(h->dbg_buf, 0, 8 * (64 + 4 * 4096)));
  return NL_OK;
}

int nl_debug_occupancy(int32_t out[8]) {
  if (!out) return NL_ERR_ARG;
  int v = 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return NL_ERR_NO_DEVICE;
  out[0] = (int32_t)(prop.maxSharedMemoryPerMultiProcessor / 1024);
  out[1] = (int32_t)(prop.sharedMemPerBlock / 1024);
  out[2] = out[3] = 0;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_sweep_count_f32<false, false>, SWEEP_WAVES * WAVE, 0);
  out[4] = v;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, k_sweep<float, MODE_FILL>, SWEEP_WAVES * WAVE, 0);
  out[5] = v;
  hipFuncAttributes fa;
  (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_sweep_count_masks_f32<false, false>));
  out[6] = (int32_t)fa.sharedSizeBytes;
  out[7] = fa.numRegs;
  return NL_OK;
}

int nl_get_build_info(nl_handle_t h, int32_t info[8]) {
  if (!h || !info) return NL_ERR_ARG;
  for (int k = 4; k < 8; k++) info[k] = 0;
  info[4] = h->b_wide ? 64 : 32;
  info[5] = h->b_mask_nb;
  info[6] = h->b_rows ? 1 + h->b_rows_v : 0;  // fine-row search: the cell table of nl_get_sorted is the fine-row table
  info[7] = h->b_use_masks && !h->b_rows && h->b_mask_nb == 1 && h->dtype == NL_F32 && !h->b_pbc && h->b_lean_small ? 1 : 0;
  info[0] = h->b_use_masks ? 1 : 0;
  info[1] = h->sweep_variant;
  info[2] = h->dtype == NL_F32 ? SweepCfg<float>::CAP : SweepCfg<double>::CAP;
  info[3] = h->num_cus;
  return NL_OK;
}

int nl_last_error(nl_handle_t h) { return h ? h->last_error : NL_ERR_ARG; }
int nl_last_hip_error(nl_handle_t h) { return h ? h->last_hip : 0; }

int nl_profile_last_build(nl_handle_t h, int32_t reps, double ms[NL_NUM_STAGES]) {
  if (!h || !ms || reps <= 0) return fail(h, NL_ERR_ARG);
  int rc = nl_synchronize(h);  // the build being profiled must have succeeded (buffers sized, list large enough)
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device));
  for (int k = 0; k < NL_NUM_STAGES; k++) ms[k] = 0;
  hipStream_t s = h->own_stream;
  HIPCHK(h, hipStreamSynchronize(h->last_stream));
  for (int r = 0; r < reps; r++) {
    rc = dispatch_build(h, h->b_q, h->b_stride, h->b_gid, h->n_rows, h->n, h->b_zlo, h->b_mzl, h->b_slab, s, h->ev);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(s));
    for (int k = 0; k < NL_STAGE_TOTAL; k++) {
      float t = 0;
      HIPCHK(h, hipEventElapsedTime(&t, h->ev[k], h->ev[k + 1]));
      ms[k] += t;
    }
    float t = 0;
    HIPCHK(h, hipEventElapsedTime(&t, h->ev[0], h->ev[NL_STAGE_TOTAL]));
    ms[NL_STAGE_TOTAL] += t;
  }
  for (int k = 0; k < NL_NUM_STAGES; k++) ms[k] /= reps;
  rc = enqueue_result_copy(h, s);
  if (rc) return rc;
  h->last_stream = s;
  h->pending = true;
  return finish(h, false);
}

int nl_profile_stages(nl_handle_t h, const void* q_dev, int32_t q_stride, int32_t n, int32_t reps,
                      double ms[NL_NUM_STAGES]) {
  if (!h || !ms || reps <= 0) return fail(h, NL_ERR_ARG);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipDeviceSynchronize());  // diagnostic entry point: whatever produced q, on whichever stream, is done
  int rc = nl_make_list(h, q_dev, q_stride, n, nullptr, 1);
  if (rc) return rc;
  return nl_profile_last_build(h, reps, ms);
}

/* ------------------------------------------------------------------ buffers */

int nl_buf_alloc(void** dev, void** host, size_t bytes) {
  if (!dev && !host) return NL_ERR_ARG;
  if (dev) *dev = nullptr;
  if (host) *host = nullptr;
  if (dev && hipMalloc(dev, bytes ? bytes : 16) != hipSuccess) return NL_ERR_NOMEM;
  if (host && hipHostMalloc(host, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
    if (dev) {
      (void)hipFree(*dev);
      *dev = nullptr;
    }
    return NL_ERR_NOMEM;
  }
  return NL_OK;
}
int nl_buf_free(void* dev, void* host) {
  int rc = NL_OK;
  if (dev && hipFree(dev) != hipSuccess) rc = NL_ERR_HIP;
  if (host && hipHostFree(host) != hipSuccess) rc = NL_ERR_HIP;
  return rc;
}
int nl_buf_h2d(void* dev, const void* host, size_t bytes) {
  return hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice) == hipSuccess ? NL_OK : NL_ERR_HIP;
}
int nl_buf_d2h(void* host, const void* dev, size_t bytes) {
  return hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost) == hipSuccess ? NL_OK : NL_ERR_HIP;
}
int nl_buf_fill32(void* dev, uint32_t pattern, size_t count) {
  if (hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(dev), (int)pattern, count) != hipSuccess) return NL_ERR_HIP;
  return hipDeviceSynchronize() == hipSuccess ? NL_OK : NL_ERR_HIP;
}
int nl_buf_fill64(void* dev, uint64_t pattern, size_t count) {
  // two interleaved 32-bit patterns: fill as 32-bit words when both halves agree, else through a staging copy
  const uint32_t lo = (uint32_t)pattern, hi = (uint32_t)(pattern >> 32);
  if (lo == hi) return nl_buf_fill32(dev, lo, count * 2);
  uint64_t* tmp = static_cast<uint64_t*>(malloc(sizeof(uint64_t) * (count ? count : 1)));
  if (!tmp) return NL_ERR_NOMEM;
  for (size_t i = 0; i < count; i++) tmp[i] = pattern;
  const hipError_t e = hipMemcpy(dev, tmp, sizeof(uint64_t) * count, hipMemcpyHostToDevice);
  free(tmp);
  return e == hipSuccess ? NL_OK : NL_ERR_HIP;
}
int nl_device_synchronize(void) { return hipDeviceSynchronize() == hipSuccess ? NL_OK : NL_ERR_HIP; }

}  // extern "C"

#include "nl_transpose.inc"
#include "nl_consumer.inc"
#include "nl_dist.inc"
