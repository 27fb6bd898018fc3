/*
 * nl_hip.h -- C ABI of libnl_hip.so: the MI355X (gfx950) Verlet neighbour-list builder.
 *
 * This is the drop-in boundary for the ONE hot path of kohnakagawa/md_neighbor_list:
 *   cell hash -> sort by cell -> 27-cell pair search with cut-off test -> compaction into the pair list.
 * The reference has no FFI layer; its boundary is the C++ class surface its two harnesses touch
 * (SURVEY.md section 8b).  Every entry point below names the reference interface it replaces
 * (file:line in the reference tree).  The header-only C++ shims include/neighlist_gpu.hpp (GPU harness
 * surface: cuda_ptr<T>, NeighListGPU<Vec,Dtype>) and include/neighlist_cpu.hpp (CPU harness surface:
 * NeighList<Vec>) are built on nothing but these functions.
 *
 * Conventions
 *   - plain C types only; every function returns an nl_status (0 = NL_OK) and never aborts or throws
 *     (the reference aborts through checkCudaErrors / std::exit(1), device_util.cuh:41-54).
 *   - a handle owns every device buffer it hands out; returned device pointers stay valid until the next
 *     nl_make_list* / nl_initialize / nl_destroy on that handle (the reference's accessors return references to
 *     members, neighlist_gpu.hpp:468-482).  The caller owns the position buffer.
 *   - one handle per device per host thread; no global state (the reference keeps function-local statics,
 *     neighlist_gpu.hpp:303, kernel_impl.cuh:222-226, and is not re-entrant).
 *   - the result contract is the reference's SCALAR CPU class (neighlist_cpu.hpp): the half list
 *     {(i,j): i<j, r2 <= rc2} with r2 = (dx*dx + dy*dy) + dz*dz evaluated without FMA in the position type and
 *     compared against a double rc2 (neighlist_cpu.hpp:215-223), over the cell pairs that class visits.
 */
#ifndef NL_HIP_H
#define NL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nl_handle_s* nl_handle_t;

typedef enum nl_dtype {
  NL_F32 = 0, /* Vec = float4-like  {x,y,z,w}, 16 B (make_list.cu:10-11) */
  NL_F64 = 1  /* Vec = double4-like {x,y,z,w}, 32 B (make_list.cu:7-8)   */
} nl_dtype;

typedef enum nl_status {
  NL_OK = 0,
  NL_ERR_ARG = 1,            /* null/negative/inconsistent argument                                            */
  NL_ERR_NOMEM = 2,          /* host or device allocation failed                                               */
  NL_ERR_OUT_OF_BOX = 3,     /* a coordinate is NaN or more than one box length outside [0,L): the reference
                                indexes out of bounds there (GenHash + one ApplyPBC wrap, neighlist_cpu.hpp:51-66) */
  NL_ERR_CAPACITY = 4,       /* pair list larger than the capacity set for this handle (async builds only;
                                the reference silently overruns MAX_PARTNERS*N, neighlist_cpu.hpp:37,76-78)    */
  NL_ERR_HIP = 5,            /* a HIP runtime call failed; see nl_last_hip_error                               */
  NL_ERR_STATE = 6,          /* call order violated (e.g. make_list before initialize, getter before a build)  */
  NL_ERR_MESH = 7,           /* fewer than 3 cells along an axis: the reference visits cell pairs twice there
                                and emits duplicate pairs; this library refuses such boxes                     */
  NL_ERR_INDEX_OVERFLOW = 8, /* the list has more than INT32_MAX entries and a 32-bit key_pointer was asked for:
                                nl_get_*_csr / the transposed list after a wide build, or a build with
                                nl_set_offset_width(32); the reference wraps silently there (neighlist_cpu.hpp:15,29) */
  NL_ERR_NO_DEVICE = 9,      /* no usable gfx950 device / wrong code object                                    */
  NL_ERR_DOMAIN = 10,        /* slab builds: a row particle outside the owned cells or a ghost inside them     */
  NL_ERR_COMM = 11           /* distributed builds: RCCL not loadable / a communicator call or the transport failed */
} nl_status;

const char* nl_status_string(int status);

/* ------------------------------------------------------------------------------------------------ lifecycle */

/* Replaces the constructors NeighListGPU(rc,Lx,Ly,Lz) (neighlist_gpu.hpp:236-255) and NeighList(rc,Lx,Ly,Lz)
 * (neighlist_cpu.hpp:380-395): mesh_size[d] = (int)(L_d / rc), ms = L/mesh_size, rc2 = rc*rc in double; ms and
 * 1/ms are rounded to `dtype` exactly as the CPU class stores them in a Vec (neighlist_cpu.hpp:12,389-391,409-411).
 * device_id < 0 selects the current HIP device. */
int nl_create(nl_handle_t* out, int dtype, double rc, double Lx, double Ly, double Lz, int device_id);

/* Replaces Initialize(N) (neighlist_gpu.hpp:268-287, neighlist_cpu.hpp:408-415): allocates every per-particle and
 * per-cell buffer for up to n_max particles.  The pair-list capacity defaults to an estimate from the number
 * density (1.3 x the ideal-gas half count + slack); see nl_set_capacity. Call once, or again to grow. */
int nl_initialize(nl_handle_t h, int32_t n_max);

/* Pair-list capacity in entries (int32 each).  A synchronous build grows the list by itself; an asynchronous
 * one (sync = 0) cannot, and reports NL_ERR_CAPACITY at the next synchronising call instead of overrunning. */
int nl_set_capacity(nl_handle_t h, int64_t max_pairs);

/* Which list the builds of this handle produce.  NL_LIST_HALF (default): the scalar CPU class's contract, every
 * pair once, in the row of min(i, j) (neighlist_cpu.hpp:233-236).  NL_LIST_FULL: the GPU kernels' contract, every
 * pair in both rows (kernel_impl.cuh:24-33: every j != i within the cut-off), as a CSR in original particle order;
 * nl_get_full_transposed then turns it into the GPU class's list[k*N + i] with coalesced writes instead of
 * deriving it from the half list.  Takes effect at the next build; the list capacity is counted in entries
 * (a full list has twice as many) and is re-estimated unless it was set by nl_set_capacity. */
enum nl_list_kind { NL_LIST_HALF = 0, NL_LIST_FULL = 1 };
int nl_set_list_kind(nl_handle_t h, int kind);

/* Width of the list offsets (key_pointer).  The reference's key_pointer_ and number_of_pairs_ are int32
 * (neighlist_cpu.hpp:15,29; neighlist_gpu.hpp:484-487) and wrap beyond INT32_MAX pairs; BASELINE config 4 has 2.5e9.
 * 0 (default): a build uses 64-bit offsets as soon as the list capacity of the handle exceeds INT32_MAX entries -- which
 * the default capacity estimate does for such boxes, and which a synchronous build that overflows reaches by growing
 * the list -- and 32-bit offsets otherwise; 32 / 64 force one width (a 32-bit build of a longer list fails with
 * NL_ERR_INDEX_OVERFLOW).  Whatever the build used, nl_get_*_csr returns int32 offsets (converted once per build if
 * needed, NL_ERR_INDEX_OVERFLOW if they cannot hold the list) and nl_get_*_csr64 int64 offsets.  NL_OFFSET_WIDTH in
 * the environment sets the default. */
int nl_set_offset_width(nl_handle_t h, int bits);

/* Launch mode of asynchronous builds ("capture launch-bound inner loops in hipGraphs").  on != 0:
 * a build is captured once into a hipGraph (memset, the kernels, the 80-byte result copy) and replayed on the caller's
 * stream by later builds with the same arguments; any change of an argument, of the list kind / periodic mode or of a
 * buffer (growth, nl_initialize) captures again.  Pays on small systems, where a build is ~11 dependent launches
 * (N = 4096: see DESIGN.md); off by default.  Also NL_GRAPH=1 in the environment. */
int nl_set_graph(nl_handle_t h, int on);

/* Distances across the periodic faces.  0 (default) = the reference: the 27-cell stencil wraps cell indices but the
 * distance is taken between the coordinates as given (neighlist_cpu.hpp:107-132,219-223), i.e. an open box.
 * 1 = minimum image (not in the reference): a stencil cell reached through a periodic face
 * is tested at its image, dx = (x_j -+ L) - x_i with the shifted coordinate rounded to the position type first; a
 * particle whose cell index was wrapped (coordinate outside [0, L), or rounding up to the box edge) is itself taken
 * at its image next to that cell.  The pair is decided once, by the row that stores it (the smaller id); with
 * NL_LIST_FULL both rows decide on their own and may differ for a pair within one ulp of the cut-off across a face.
 * Slab builds: the two ghost layers of the box-end ranks are the periodic images (the caller sends the layers
 * unshifted, as for the open box).  Takes effect at the next build. */
int nl_set_periodic(nl_handle_t h, int minimum_image);

int nl_destroy(nl_handle_t h);

/* --------------------------------------------------------------------------------------------------- build */

/* Replaces MakeNeighList(q, N, sync, tblock_size, smem_hei) (neighlist_gpu.hpp:289-466) and MakeNeighList(q, N)
 * (neighlist_cpu.hpp:417-435).  q_dev: device pointer to n positions, `q_stride` scalars apart (4 for the
 * float4/double4 Vec of make_list.cu, 3 for the {x,y,z} Vec of make_list.cpp:26-32).  Positions are read, never
 * reordered (the reference's SortPtclData is commented out, neighlist_cpu.hpp:421).  stream: the hipStream_t the
 * build is enqueued on; NULL is HIP's null (default) stream, where the reference launches (make_list.cu:124-127).
 * The build is ordered after everything already queued on that stream (the kernel that produced q, a halo
 * exchange) and nothing else: positions written on ANOTHER stream must be fenced by the caller.  sync != 0 waits for the build and returns its status; sync == 0 only
 * enqueues (the reference's timing loop, make_list.cu:124-127) and errors surface at the next nl_synchronize /
 * getter.  tblock_size and smem_hei of the reference select among its CUDA variants and have no counterpart. */
int nl_make_list(nl_handle_t h, const void* q_dev, int32_t q_stride, int32_t n, void* stream, int sync);

/* Slab (domain-decomposed) build, SURVEY.md section 8e -- no reference counterpart (the reference is single GPU).
 * The handle describes the GLOBAL box.  This rank owns the cell layers z in [z_lo, z_hi) of the global mesh and
 * passes n_rows owned particles first, then n - n_rows ghost particles lying in the two periodic neighbour layers
 * (z_lo-1 and z_hi, modulo mesh_z).  gid_dev: global particle ids (NULL = identity; NL_GID_IN_W = the id is stored
 * in the w component of each position, as the bit pattern of an int32 (F32) / int64 (F64), which halves the
 * number of halo messages).  Rows are built for the
 * owned particles only: row r holds the global ids j > gid[r] within the cut-off, so that the union over ranks
 * is exactly the global half list.  z_lo = 0, z_hi = mesh_z, n_rows = n is the single-GPU build. */
#define NL_GID_IN_W ((const int32_t*)1)
int nl_make_list_slab(nl_handle_t h, const void* q_dev, int32_t q_stride, const int32_t* gid_dev, int32_t n_rows,
                      int32_t n, int32_t z_lo, int32_t z_hi, void* stream, int sync);

/* The same build in two calls, so that the halo exchange overlaps its first part: _begin enqueues what needs only
 * the OWNED particles q[0, n_rows) -- the binning of the owned layers, about a tenth of the build -- and may be called
 * while the ghost rows q[n_rows, n) are still being received; _finish (same stream, after the caller has made that
 * stream wait for the exchange) enqueues the rest: the binning of the ghosts, the search, the scan, the expansion.
 * n_ghost_lo = how many of the ghosts lie in the lower neighbour layer (z_lo - 1): the owned particles are placed
 * behind them in the cell-sorted array before any ghost has been seen, so the number is part of the call; it is
 * verified when the ghosts are binned (NL_ERR_DOMAIN).  The result is the one of nl_make_list_slab.  Builds that
 * have nothing to overlap (single rank; NL_BINNING=1) do all their work in _finish. */
int nl_make_list_slab_begin(nl_handle_t h, const void* q_dev, int32_t q_stride, const int32_t* gid_dev, int32_t n_rows,
                            int32_t n, int32_t n_ghost_lo, int32_t z_lo, int32_t z_hi, void* stream);
int nl_make_list_slab_finish(nl_handle_t h, void* stream, int sync);

/* ------------------------------------------------------------------------------- the decomposed build, whole */

/* The domain-decomposed build with its halo exchange inside the library (no reference
 * counterpart: the reference is single GPU, make_list.cu:122-127).  One process per GPU; rank r of `world` owns the
 * cell layers nl_comm_layers returns for it (contiguous runs of the global mesh's z layers, as even as possible) and
 * passes its owned particles only; per build the library packs the two boundary layers, tells the two z-neighbours how
 * many particles come (the counts change from build to build in a moving system), moves the layers into the ghost rows
 * of the caller's buffer and runs nl_make_list_slab on owned + ghosts.  Point-to-point only, no collective.
 *
 * nl_comm_create: RCCL over xGMI.  unique_id = the NL_UNIQUE_ID_BYTES bytes rank 0 got from nl_comm_unique_id and handed
 *   to every rank out of band (ncclGetUniqueId / ncclCommInitRank).  RCCL is resolved at run time (dlopen: the copy the
 *   process already uses, e.g. PyTorch's, else /opt/rocm/lib): NL_ERR_COMM if it cannot be loaded.  The transfer runs on
 *   a communication stream under the binning of the owned layers (nl_make_list_slab_begin / _finish).
 * nl_comm_create_callbacks: the caller supplies the transport, a blocking host-memory exchange
 *   fn(user, peer_to, send, send_bytes, peer_from, recv, recv_bytes) -> 0 on success, called twice per message round in
 *   the same order on every rank (first everybody sends to rank-1 and receives from rank+1, then the other way; a zero
 *   byte count means that side is skipped).  For MPI / gloo / test harnesses; the layers are staged through pinned host memory.
 * nl_make_list_distributed: q_dev = this rank's positions {x, y, z, w} (stride 4) with the GLOBAL particle id in w
 *   (bit pattern of an int32 for NL_F32, of an int64 for NL_F64: NL_GID_IN_W), n_owned rows filled by the caller and room
 *   for q_capacity rows: the ghosts are written behind the owned rows.  Every owned particle must lie in the rank's
 *   layers (NL_ERR_DOMAIN otherwise: migrating particles between ranks is the caller's job); NL_ERR_CAPACITY when
 *   owned + ghosts exceed q_capacity or the handle's n_max.  Rows (nl_get_half_csr ...) are those of the owned
 *   particles and hold global ids, as after nl_make_list_slab.  With sync == 0 a steady-state build returns without
 *   waiting for the device: the boundary layers travel in messages of negotiated capacity with their counts in a header,
 *   the ghost counts stay on the device (nl_dist.inc).  The first build of a communicator, and the build after one whose
 *   layer outgrew its message (that one reports NL_ERR_CAPACITY at its synchronisation; with sync == 1 it renegotiates and
 *   repeats itself), exchange the counts through the host.
 * nl_distributed_ghosts: the ghost counts of the last build (rows [n_owned, n_owned + lo) and the hi rows behind); waits for
 *   that build. */
typedef struct nl_comm_s* nl_comm_t;
#define NL_UNIQUE_ID_BYTES 128
typedef int (*nl_sendrecv_fn)(void* user, int peer_to, const void* send, size_t send_bytes, int peer_from, void* recv,
                              size_t recv_bytes);
int nl_comm_unique_id(void* id_out /* NL_UNIQUE_ID_BYTES */);
int nl_comm_create(nl_comm_t* out, int rank, int world, const void* unique_id, int device_id);
int nl_comm_create_callbacks(nl_comm_t* out, int rank, int world, nl_sendrecv_fn fn, void* user, int device_id);
int nl_comm_destroy(nl_comm_t comm);
int nl_comm_layers(nl_handle_t h, nl_comm_t comm, int32_t* z_lo, int32_t* z_hi);
int nl_make_list_distributed(nl_handle_t h, nl_comm_t comm, void* q_dev, int32_t q_capacity, int32_t n_owned, void* stream,
                             int sync);
int nl_distributed_ghosts(nl_comm_t comm, int32_t* n_ghost_lo, int32_t* n_ghost_hi);

/* Waits for the last enqueued build and returns its status (replaces the harness's
 * checkCudaErrors(cudaDeviceSynchronize()), make_list.cu:128). */
int nl_synchronize(nl_handle_t h);

/* ------------------------------------------------------------------------------------------------- results */

/* The CPU class's accessors (neighlist_cpu.hpp:437-463): key_pointer()[N+1], sorted_list()[P],
 * number_of_partners()[N] (half counts, on min(i,j)), number_of_pairs() = P -- as DEVICE pointers.
 * Partners of particle i are sorted_list[key_pointer[i] .. key_pointer[i+1]), in no particular order (the
 * reference's order is its visit order; its own check sorts each segment first, make_list.cpp:120-128,211).
 * Synchronises. Any out pointer may be NULL. */
int nl_get_half_csr(nl_handle_t h, const int32_t** key_pointer_dev, const int32_t** sorted_list_dev,
                    const int32_t** number_of_partners_dev, int64_t* npairs);

/* The same arrays of a NL_LIST_FULL build: key_pointer[N+1], list[2P], full counts [N]; *nentries = 2P.
 * NL_ERR_STATE if the last build was a half build (and nl_get_half_csr after a full build). */
int nl_get_full_csr(nl_handle_t h, const int32_t** key_pointer_dev, const int32_t** list_dev,
                    const int32_t** number_of_partners_dev, int64_t* nentries);

/* The same lists with 64-bit offsets: key_pointer[N+1] as int64 (no reference counterpart: its offsets are int32,
 * neighlist_cpu.hpp:29, and BASELINE config 4's 2.5e9 pairs do not fit them). */
int nl_get_half_csr64(nl_handle_t h, const int64_t** key_pointer_dev, const int32_t** sorted_list_dev,
                      const int32_t** number_of_partners_dev, int64_t* npairs);
int nl_get_full_csr64(nl_handle_t h, const int64_t** key_pointer_dev, const int32_t** list_dev,
                      const int32_t** number_of_partners_dev, int64_t* nentries);

/* Order-independent checksum of the list of the last build, computed on the device: the wrapping sum over entries
 * (row i, partner j) of mix((id_i << 32) | j), mix(v): v *= 0x9E3779B97F4A7C15; v ^= v >> 29 -- the pair-set hash of
 * the reference harness's known answers -- with id_i the row's global id (slab builds: the ids the
 * build was given).  The sum over the ranks of a decomposed build is the checksum of the global list.  Synchronises. */
int nl_list_checksum(nl_handle_t h, uint64_t* checksum, int64_t* nentries);

/* The GPU class's accessors neigh_list() / number_of_partners() (neighlist_gpu.hpp:468-482): the FULL list in
 * the transposed layout list[k * row_stride + i], k < count[i], original particle ids: converted from the full
 * CSR of a NL_LIST_FULL build (one coalesced pass), or derived on the device from the half list of a NL_LIST_HALF
 * build (scattered writes: about ten times slower).  Entries k >= count[i] hold -1 when the buffer is first
 * allocated or grown and whatever an earlier build left there afterwards (the reference fills -1 once in
 * Initialize and never again, neighlist_gpu.hpp:271).  *max_partners receives max_i count[i].  Synchronises. */
int nl_get_full_transposed(nl_handle_t h, const int32_t** list_dev, const int32_t** count_dev, int64_t* row_stride,
                           int32_t* max_partners);

/* number_of_pairs(): P, the half-pair count (neighlist_cpu.hpp:437-439).  The GPU class returns the sum of the
 * full counts = 2P (neighlist_gpu.hpp:484-487); the C++ shim doubles it.  Synchronises. */
int nl_number_of_pairs(nl_handle_t h, int64_t* npairs);

/* ------------------------------------------------------------------------------------- periodic re-sorting */

/* The physical re-sort the reference declares and never performs: SORT_FREQ = 50 (neighlist_gpu.hpp:72), CopyGather
 * (neighlist_gpu.hpp:144-151), SortPtclData (neighlist_cpu.hpp:176-180, its call commented out at :421).  A build
 * already sorts a COPY of the positions into cell order; an MD loop that permutes its own per-particle arrays the same
 * way every SORT_FREQ builds keeps them spatially coherent, which makes the binning pass of the next builds and every
 * gather through the list (forces) faster.
 *   nl_get_cell_order: order[s] = input index of the particle the last build placed at cell-ordered slot s
 *     (the reference's ptcl_id_in_mesh, neighlist_gpu.hpp:153-199), a device pointer valid until the next build.
 *   nl_resort: array[s] <- array[order[s]] in place for one per-particle array of elem_bytes (4, 8, 12, 16, 24 or 32)
 *     per particle: positions, velocities, ids, ... -- call it once per array, then rebuild: the new list is the list
 *     of the permuted particles (indices are the NEW positions in the arrays).  Enqueued on `stream`; single-device
 *     builds only.  Synchronises with the last build first. */
int nl_get_cell_order(nl_handle_t h, const int32_t** order_dev, int32_t* n);
int nl_resort(nl_handle_t h, void* array_dev, size_t elem_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------- a consumer */

/* Truncated Lennard-Jones forces from the list of the last build (the reference stops at
 * the list: its momentum array is allocated and never used, make_list.cpp:135,138-140).  q_dev: the positions the
 * list was built from (or moved by less than the skin), same dtype and stride; f_dev: n x 4 values of that dtype,
 * {fx, fy, fz, pe_i} with pe_i = half the pair energies of particle i; pairs beyond rc_force (<= the list's cut-off)
 * are skipped; distances are taken as the list took them: between the coordinates as given (open box, the reference's
 * rule) or, after nl_set_periodic(1), at the minimum image.  After a NL_LIST_FULL build every row gathers its partners and
 * writes its force once; after a NL_LIST_HALF build every pair is evaluated once and the reaction is added to the
 * partner with floating-point atomics (f_dev is zeroed first).  Enqueued on `stream` (NULL = the null stream);
 * waits for the build first. */
int nl_lj_forces(nl_handle_t h, const void* q_dev, int32_t q_stride, double epsilon, double sigma, double rc_force,
                 void* f_dev, void* stream);

/* ------------------------------------------------------------------------------------------- introspection */

int nl_get_mesh(nl_handle_t h, int32_t mesh[3], int64_t* ncell);
/* Cell-sorted state of the last build, for tests of the hash/sort stage (a3-a5 of SURVEY.md section 8):
 * cell_start[ncell_local + 1]; sorted positions {x,y,z,id} (16 B for F32, 32 B {double x,y,z; int32 id,row}
 * for F64); sorted_row[n] = input index of each sorted slot.  After a fine-row build (nl_get_build_info, info[6] != 0)
 * the cell's particles are additionally ordered by the quarter of the cell they lie in along z and the table has
 * 4 * ncell_local + 1 entries: entry (cell * 4 + quarter) = first slot of that quarter; every fourth entry is the
 * cell_start of the plain build. */
int nl_get_sorted(nl_handle_t h, const int32_t** cell_start_dev, const void** sorted_pos_dev,
                  const int32_t** sorted_row_dev, int64_t* ncell_local);
/* Diagnostic cycle accumulators of the kernels (filled only when NL_DEBUG_FLAGS & 4 is set in the environment). */
int nl_debug_read(nl_handle_t h, uint64_t* out, int32_t n, int reset);
int nl_debug_occupancy(int32_t out[8]); /* LDS per CU/block (KiB), occupancy API answers, LDS bytes, registers */
/* How the last build was organised: info[0] = 1 when the COUNT sweep kept hit masks and the list was expanded from
 * them (0: two distance sweeps), info[1] = configured sweep variant, info[2] = LDS batch capacity (particles),
 * info[3] = compute units of the device, info[4] = width of the list offsets the build used (32 / 64), info[5] = mask rows per particle (1; up to 7 in a
 * dense build: one per LDS batch of the stencil stream), info[6] = 0, or 1 + c when the build took the fine-row search
 * (nl_rows.hpp; c = its LDS configuration 0..2) -- the table nl_get_sorted returns is then the fine-row table,
 * info[7] = 1 when the build used the small instances of the COUNT sweep and the expansion (sparse boxes: 2 waves /
 * 1 wave per cell, half the LDS buffer), else 0. */
int nl_get_build_info(nl_handle_t h, int32_t info[8]);
int nl_last_error(nl_handle_t h);     /* status of the last failed call on this handle */
int nl_last_hip_error(nl_handle_t h); /* raw hipError_t behind the last NL_ERR_HIP */

/* Per-kernel device time of one build, measured with HIP events on the stream the kernels were launched on.
 * Runs `reps` builds of the given positions and returns average milliseconds per build for each stage
 * (NL_STAGE_*) in ms[NL_NUM_STAGES]; ms[NL_STAGE_TOTAL] is first-launch to last-kernel-end. */
enum {
  NL_STAGE_HASH = 0,     /* cell hash + per-cell rank                       */
  NL_STAGE_CELL_SCAN = 1,/* exclusive scan of the cell histogram            */
  NL_STAGE_REORDER = 2,  /* physical reorder of positions into cell order   */
  NL_STAGE_COUNT = 3,    /* pair search, counting pass                      */
  NL_STAGE_ROW_SCAN = 4, /* exclusive scan of the counts -> key_pointer     */
  NL_STAGE_FILL = 5,     /* pair search, list-filling pass                  */
  NL_STAGE_TOTAL = 6,
  NL_NUM_STAGES = 7
};
int nl_profile_stages(nl_handle_t h, const void* q_dev, int32_t q_stride, int32_t n, int32_t reps,
                      double ms[NL_NUM_STAGES]);
/* Same, re-running the last successful build (single-device or slab) with the arguments it was given; the
 * position / id buffers of that build must still be alive. */
int nl_profile_last_build(nl_handle_t h, int32_t reps, double ms[NL_NUM_STAGES]);

/* --------------------------------------------------------------------------------- buffers (cuda_ptr shim) */

/* Back the reference's cuda_ptr<T> (cuda_ptr.cuh:11-112): a device buffer paired with a pinned host buffer. */
int nl_buf_alloc(void** dev, void** host, size_t bytes);          /* allocate(), cuda_ptr.cuh:40-45; either
                                                                     pointer may be NULL (that half is skipped) */
int nl_buf_free(void* dev, void* host);                           /* deallocate(), cuda_ptr.cuh:107-111    */
int nl_buf_h2d(void* dev, const void* host, size_t bytes);        /* host2dev(), cuda_ptr.cuh:47-53        */
int nl_buf_d2h(void* host, const void* dev, size_t bytes);        /* dev2host(), cuda_ptr.cuh:62-69        */
int nl_buf_fill32(void* dev, uint32_t pattern, size_t count);     /* set_val() device half, cuda_ptr.cuh:79-89 */
int nl_buf_fill64(void* dev, uint64_t pattern, size_t count);
int nl_device_synchronize(void);                                  /* make_list.cu:128                      */
int nl_device_count(int* count);

#ifdef __cplusplus
}
#endif
#endif /* NL_HIP_H */
