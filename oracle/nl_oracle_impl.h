/*
 * oracle/nl_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see nl_oracle.c header).
 *
 * Type-generic body of the CPU restatement.  Included twice by nl_oracle.c with
 *   REAL   = float | double      (the scalar type of the reference's `Vec`)
 *   SUF(x) = x##_f32 | x##_f64
 *
 * Every function names the reference lines it restates (/root/reference/...).
 * Arithmetic is written so that a C compiler with -ffp-contract=off evaluates it
 * exactly like the reference's C++ does for Vec{REAL x,y,z}:
 *   products/sums of REAL stay in REAL; the cut-off comparison promotes r2 to double
 *   because search_length2_ is a double (neighlist_cpu.hpp:13,223).
 */

/* Grid parameters: ctor neighlist_cpu.hpp:380-395 and Initialize :408-411.
 * ms_/ims_ live in a Vec, i.e. they are rounded to REAL (:12). */
typedef struct {
  int32_t m[3];
  int64_t ncell;
  REAL ms[3], ims[3];
  double rc2;
} SUF(grid);

static int SUF(grid_init)(SUF(grid) * g, double rc, double Lx, double Ly, double Lz) {
  const double L[3] = {Lx, Ly, Lz};
  for (int d = 0; d < 3; d++) {
    g->m[d] = (int32_t)(L[d] / rc);              /* :384-386 */
    if (g->m[d] <= 0) return NLO_ERR_ARG;
    g->ms[d] = (REAL)(L[d] / g->m[d]);           /* :389-391 (double quotient stored into Vec) */
    g->ims[d] = (REAL)(1.0 / g->ms[d]);          /* :409-411 (1.0 is double; ms_ promoted) */
  }
  g->ncell = (int64_t)g->m[0] * g->m[1] * g->m[2]; /* :387 */
  g->rc2 = rc * rc;                              /* :394 */
  return NLO_OK;
}

/* ApplyPBC neighlist_cpu.hpp:61-66: a single +-m wrap, nothing more. */
static void SUF(apply_pbc)(const SUF(grid) * g, int32_t* idx) {
  for (int d = 0; d < 3; d++) {
    if (idx[d] < 0) idx[d] += g->m[d];
    if (idx[d] >= g->m[d]) idx[d] -= g->m[d];
  }
}

/* GenHash(idx) neighlist_cpu.hpp:42-49. */
static int64_t SUF(hash_idx)(const SUF(grid) * g, const int32_t* idx) {
  return idx[0] + ((int64_t)idx[1] + (int64_t)idx[2] * g->m[1]) * g->m[0];
}

/* GenHash(q) neighlist_cpu.hpp:51-59.  Returns -1 where the reference would index
 * out of bounds (coordinate further than one box length outside [0,L)). */
static int64_t SUF(hash_pos)(const SUF(grid) * g, const REAL* q) {
  int32_t idx[3];
  for (int d = 0; d < 3; d++) {
    const REAL t = q[d] * g->ims[d];
    if (!(t > (REAL)-2147483000.0 && t < (REAL)2147483000.0)) return -1; /* NaN/overflow: UB in the reference */
    idx[d] = (int32_t)t;
  }
  SUF(apply_pbc)(g, idx);
  for (int d = 0; d < 3; d++)
    if (idx[d] < 0 || idx[d] >= g->m[d]) return -1;
  return SUF(hash_idx)(g, idx);
}

/* RegistInteractPair's accept rule neighlist_cpu.hpp:215-223:
 * d = qj - qi per component, r2 = dx*dx + dy*dy + dz*dz (left to right), reject iff r2 > rc2(double). */
static inline int SUF(accept)(const REAL* qi, const REAL* qj, double rc2) {
  const REAL dx = qj[0] - qi[0];
  const REAL dy = qj[1] - qi[1];
  const REAL dz = qj[2] - qi[2];
  const REAL r2 = dx * dx + dy * dy + dz * dz;
  return !((double)r2 > rc2);
}

/*
 * The whole build: MakeNeighList neighlist_cpu.hpp:417-435 with the
 * WITHOUT_LOOP_FUSION pair loop (:239-270; the other two variants visit the same
 * pair set in another order), then MakeNeighListForEachPtcl (:361-377).
 *
 * q: N particles, `stride` REALs apart, x,y,z first.
 * Outputs (caller frees *sorted_list with nl_oracle_free):
 *   number_of_partners[N]  half counts, counted on min(i,j)      (:235)
 *   key_pointer[N+1]       exclusive prefix sum, 64-bit here      (:362-367)
 *   *sorted_list[P]        partners in the reference's visit order (:369-372)
 */
int SUF(nl_oracle_build)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly,
                         double Lz, int32_t* number_of_partners, int64_t* key_pointer,
                         int32_t** sorted_list, int64_t* npairs) {
  SUF(grid) g;
  if (N < 0 || N > 2147483647LL || stride < 3) return NLO_ERR_ARG;
  int rc_ = SUF(grid_init)(&g, rc, Lx, Ly, Lz);
  if (rc_) return rc_;
  const int64_t M = g.ncell;

  int32_t* cell_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t* cell_n = (int32_t*)calloc((size_t)M, sizeof(int32_t));
  int64_t* cell_beg = (int64_t*)malloc(sizeof(int64_t) * (size_t)(M + 1));
  int64_t* cursor = (int64_t*)malloc(sizeof(int64_t) * (size_t)(M + 1));
  int32_t* id_in_cell = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int32_t* neigh = (int32_t*)malloc(sizeof(int32_t) * (size_t)M * 13);
  int64_t cap = N * 64 + 1024, P = 0;
  int32_t* key = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  int32_t* par = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  int ret = NLO_OK;
  if (!cell_of || !cell_n || !cell_beg || !cursor || !id_in_cell || !neigh || !key || !par) {
    ret = NLO_ERR_NOMEM;
    goto done;
  }

  /* MakeNeighMeshId :107-132 -- the first 13 of the 27 offsets in (jz,jy,jx) order. */
  {
    int64_t c = 0;
    for (int32_t iz = 0; iz < g.m[2]; iz++)
      for (int32_t iy = 0; iy < g.m[1]; iy++)
        for (int32_t ix = 0; ix < g.m[0]; ix++, c++) {
          int k = 0;
          for (int32_t jz = -1; jz < 2 && k < 13; jz++)
            for (int32_t jy = -1; jy < 2 && k < 13; jy++)
              for (int32_t jx = -1; jx < 2 && k < 13; jx++) {
                int32_t idx[3] = {ix + jx, iy + jy, iz + jz};
                SUF(apply_pbc)(&g, idx);
                /* With a mesh of 1 along an axis one wrap is not enough (ix+jx = -1 -> 0 ok,
                 * = 1 -> 0 ok), so every index is in range for m >= 1. */
                neigh[13 * c + k++] = (int32_t)SUF(hash_idx)(&g, idx);
              }
        }
  }

  /* MakeMeshidOfPtcl :134-144 */
  for (int64_t i = 0; i < N; i++) {
    const int64_t h = SUF(hash_pos)(&g, q + (size_t)i * stride);
    if (h < 0) {
      ret = NLO_ERR_OUT_OF_BOX;
      goto done;
    }
    cell_of[i] = (int32_t)h;
    cell_n[h]++;
  }
  /* MakeNextDest :146-165 -- stable counting sort of ids by cell */
  cell_beg[0] = cursor[0] = 0;
  for (int64_t c = 0; c < M; c++) cell_beg[c + 1] = cursor[c + 1] = cell_beg[c] + cell_n[c];
  for (int64_t i = 0; i < N; i++) id_in_cell[cursor[cell_of[i]]++] = (int32_t)i;

  /* MakePairListNaive :239-270 */
  memset(number_of_partners, 0, sizeof(int32_t) * (size_t)N);
  for (int64_t c = 0; c < M; c++) {
    const int64_t ib = cell_beg[c], ie = cell_beg[c + 1];
    for (int64_t a = ib; a < ie; a++) {
      const int32_t pi = id_in_cell[a];
      const REAL* qi = q + (size_t)pi * stride;
      for (int k = 0; k <= 13; k++) {
        int64_t jb, je;
        if (k < 13) { /* "for different mesh" :253-261 */
          const int32_t jc = neigh[13 * c + k];
          jb = cell_beg[jc];
          je = cell_beg[jc + 1];
        } else { /* "for same mesh" :264-267 */
          jb = a + 1;
          je = ie;
        }
        for (int64_t b = jb; b < je; b++) {
          const int32_t pj = id_in_cell[b];
          if (!SUF(accept)(qi, q + (size_t)pj * stride, g.rc2)) continue;
          if (P == cap) {
            cap *= 2;
            int32_t* k2 = (int32_t*)realloc(key, sizeof(int32_t) * (size_t)cap);
            int32_t* p2 = (int32_t*)realloc(par, sizeof(int32_t) * (size_t)cap);
            if (k2) key = k2;
            if (p2) par = p2;
            if (!k2 || !p2) {
              ret = NLO_ERR_NOMEM;
              goto done;
            }
          }
          const int32_t lo = pi < pj ? pi : pj, hi = pi < pj ? pj : pi; /* :225-232 */
          key[P] = lo;
          par[P] = hi;
          number_of_partners[lo]++;
          P++;
        }
      }
    }
  }

  /* MakeNeighListForEachPtcl :361-377 */
  {
    int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)(P > 0 ? P : 1));
    int64_t* kp2 = (int64_t*)malloc(sizeof(int64_t) * (size_t)(N + 1));
    if (!out || !kp2) {
      free(out);
      free(kp2);
      ret = NLO_ERR_NOMEM;
      goto done;
    }
    key_pointer[0] = kp2[0] = 0;
    for (int64_t i = 0; i < N; i++) key_pointer[i + 1] = kp2[i + 1] = key_pointer[i] + number_of_partners[i];
    for (int64_t p = 0; p < P; p++) out[kp2[key[p]]++] = par[p];
    free(kp2);
    *sorted_list = out;
    *npairs = P;
  }

done:
  free(cell_of);
  free(cell_n);
  free(cell_beg);
  free(cursor);
  free(id_in_cell);
  free(neigh);
  free(key);
  free(par);
  return ret;
}

/* Cell id of every particle, for unit-testing the device hash kernel
 * (GenHash neighlist_cpu.hpp:51-59). cell[i] = -1 where the reference is out of bounds. */
int SUF(nl_oracle_cells)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly,
                         double Lz, int32_t* cell, int32_t* mesh3) {
  SUF(grid) g;
  int rc_ = SUF(grid_init)(&g, rc, Lx, Ly, Lz);
  if (rc_) return rc_;
  for (int d = 0; d < 3; d++) mesh3[d] = g.m[d];
  for (int64_t i = 0; i < N; i++) cell[i] = (int32_t)SUF(hash_pos)(&g, q + (size_t)i * stride);
  return NLO_OK;
}

/*
 * Brute force half list, make_list.cpp:79-99 (+ make_sorted_list :101-118): for i, for j>i,
 * reject iff dr2 > SEARCH_LENGTH2.  SEARCH_LENGTH2 there has type Dtype (= the position type,
 * make_list.cpp:15,24); rc2_in_real selects that (1) or the class's double rc2 (0).
 * Output is already canonical (ascending j per i).
 */
int SUF(nl_oracle_bruteforce)(const REAL* q, int32_t stride, int64_t N, double rc, int rc2_in_real,
                              int32_t* number_of_partners, int64_t* key_pointer,
                              int32_t** sorted_list, int64_t* npairs) {
  const double rc2 = rc2_in_real ? (double)((REAL)rc * (REAL)rc) : rc * rc;
  int64_t cap = N * 64 + 1024, P = 0;
  int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  if (!out) return NLO_ERR_NOMEM;
  key_pointer[0] = 0;
  for (int64_t i = 0; i < N; i++) {
    const REAL* qi = q + (size_t)i * stride;
    int32_t n = 0;
    for (int64_t j = i + 1; j < N; j++) {
      if (!SUF(accept)(qi, q + (size_t)j * stride, rc2)) continue;
      if (P == cap) {
        cap *= 2;
        int32_t* o2 = (int32_t*)realloc(out, sizeof(int32_t) * (size_t)cap);
        if (!o2) {
          free(out);
          return NLO_ERR_NOMEM;
        }
        out = o2;
      }
      out[P++] = (int32_t)j;
      n++;
    }
    number_of_partners[i] = n;
    key_pointer[i + 1] = P;
  }
  *sorted_list = out;
  *npairs = P;
  return NLO_OK;
}

/*
 * Minimum-image half list.  NOT a restatement of the reference -- it has no such mode --
 * but the definition the HIP path's nl_set_periodic(1) is tested against.  Written independently of the device code:
 *   - the cell index of a coordinate is FLOOR(q * ims) here (GenHash truncates toward zero, which files a particle
 *     at -0.3 cells into cell 0; harmless for the reference's open box, wrong for images), wrapped once by +-m;
 *   - every particle is first taken at the image that lies in its (wrapped) cell: per axis, the coordinate + L if
 *     the index was negative, - L if it was >= m, rounded to REAL;
 *   - for particle i (cell c, smaller id) and particle j (larger id) in one of the 27 neighbour cells of c, reached
 *     through offset (jx,jy,jz): s_d = -L_d if c_d + j_d < 0, +L_d if c_d + j_d >= m_d, else 0 (L_d rounded to REAL);
 *     d = (q_j + s) - q_i with the shifted coordinate rounded to REAL first; r2 = dx*dx + dy*dy + dz*dz;
 *     the pair is kept unless (double)r2 > rc2.
 * Needs >= 3 cells per axis.  Output: canonical CSR (ascending partners), as the brute force.
 */
static int SUF(build_pbc_impl)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly, double Lz,
                               int32_t* number_of_partners, int64_t* key_pointer, int32_t** sorted_list,
                               int64_t* npairs, int full) {
  SUF(grid) g;
  if (N < 0 || N > 2147483647LL || stride < 3) return NLO_ERR_ARG;
  int rc_ = SUF(grid_init)(&g, rc, Lx, Ly, Lz);
  if (rc_) return rc_;
  if (g.m[0] < 3 || g.m[1] < 3 || g.m[2] < 3) return NLO_ERR_ARG;
  const REAL L[3] = {(REAL)Lx, (REAL)Ly, (REAL)Lz};
  const int64_t M = g.ncell;
  REAL* e = (REAL*)malloc(sizeof(REAL) * 3 * (size_t)(N > 0 ? N : 1)); /* effective coordinates */
  int32_t* cidx = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(N > 0 ? N : 1));
  int64_t* cell_beg = (int64_t*)calloc((size_t)(M + 2), sizeof(int64_t));
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int64_t cap = N * (full ? 160 : 64) + 1024, P = 0;
  int32_t* out = (int32_t*)malloc(sizeof(int32_t) * (size_t)cap);
  int ret = NLO_OK;
  if (!e || !cidx || !cell_beg || !ids || !out) {
    ret = NLO_ERR_NOMEM;
    goto done;
  }
  for (int64_t i = 0; i < N; i++) {
    const REAL* qi = q + (size_t)i * stride;
    for (int d = 0; d < 3; d++) {
      const REAL t = qi[d] * g.ims[d];
      if (!(t > (REAL)-2147483000.0 && t < (REAL)2147483000.0)) {
        ret = NLO_ERR_OUT_OF_BOX;
        goto done;
      }
      int32_t v = (int32_t)t;
      if (t < 0 && (REAL)v != t) v -= 1; /* floor */
      REAL x = qi[d];
      if (v < 0) v += g.m[d], x = qi[d] + L[d];
      if (v >= g.m[d]) v -= g.m[d], x = qi[d] - L[d];
      if (v < 0 || v >= g.m[d]) {
        ret = NLO_ERR_OUT_OF_BOX;
        goto done;
      }
      cidx[3 * i + d] = v;
      e[3 * i + d] = x;
    }
    cell_beg[SUF(hash_idx)(&g, cidx + 3 * i) + 2]++;
  }
  for (int64_t c = 0; c < M; c++) cell_beg[c + 2] += cell_beg[c + 1];
  for (int64_t i = 0; i < N; i++) ids[cell_beg[SUF(hash_idx)(&g, cidx + 3 * i) + 1]++] = (int32_t)i;
  /* now cell_beg[c] .. cell_beg[c+1] are the particles of cell c */
  key_pointer[0] = 0;
  for (int64_t i = 0; i < N; i++) {
    const int64_t row0 = P;
    for (int32_t jz = -1; jz < 2; jz++)
      for (int32_t jy = -1; jy < 2; jy++)
        for (int32_t jx = -1; jx < 2; jx++) {
          const int32_t off[3] = {jx, jy, jz};
          int32_t nc[3];
          REAL s[3];
          for (int d = 0; d < 3; d++) {
            nc[d] = cidx[3 * i + d] + off[d];
            s[d] = 0;
            if (nc[d] < 0) nc[d] += g.m[d], s[d] = -L[d];
            if (nc[d] >= g.m[d]) nc[d] -= g.m[d], s[d] = L[d];
          }
          const int64_t c = SUF(hash_idx)(&g, nc);
          for (int64_t b = cell_beg[c]; b < cell_beg[c + 1]; b++) {
            const int32_t j = ids[b];
            if (full ? j == i : j <= i) continue;
            const REAL xs = e[3 * (int64_t)j] + s[0], ys = e[3 * (int64_t)j + 1] + s[1], zs = e[3 * (int64_t)j + 2] + s[2];
            const REAL dx = xs - e[3 * i], dy = ys - e[3 * i + 1], dz = zs - e[3 * i + 2];
            const REAL r2 = dx * dx + dy * dy + dz * dz;
            if ((double)r2 > g.rc2) continue;
            if (P == cap) {
              cap *= 2;
              int32_t* o2 = (int32_t*)realloc(out, sizeof(int32_t) * (size_t)cap);
              if (!o2) {
                ret = NLO_ERR_NOMEM;
                goto done;
              }
              out = o2;
            }
            out[P++] = j;
          }
        }
    /* ascending partners */
    for (int64_t a = row0 + 1; a < P; a++) {
      const int32_t v = out[a];
      int64_t b = a - 1;
      while (b >= row0 && out[b] > v) out[b + 1] = out[b], b--;
      out[b + 1] = v;
    }
    number_of_partners[i] = (int32_t)(P - row0);
    key_pointer[i + 1] = P;
  }
  *sorted_list = out;
  out = NULL;
  *npairs = P;
done:
  free(e);
  free(cidx);
  free(cell_beg);
  free(ids);
  free(out);
  return ret;
}

int SUF(nl_oracle_build_pbc)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly, double Lz,
                             int32_t* number_of_partners, int64_t* key_pointer, int32_t** sorted_list,
                             int64_t* npairs) {
  return SUF(build_pbc_impl)(q, stride, N, rc, Lx, Ly, Lz, number_of_partners, key_pointer, sorted_list, npairs, 0);
}

/* The FULL minimum-image list (both directions): row i holds every j != i accepted IN THE FRAME OF i, i.e. with the
 * image of j taken as (q_j + s) rounded to REAL, exactly as above but for all j -- what a code with ghost particles
 * computes.  Because (q_j + s) - q_i and (q_i - s) - q_j are rounded differently, a pair whose distance is within one
 * rounding error of the cut-off can be present in one direction only: the full list is NOT defined as the
 * symmetrised half list in this mode (it is in the open box, where the two differences are exact negatives). */
int SUF(nl_oracle_build_pbc_full)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly, double Lz,
                                  int32_t* number_of_partners, int64_t* key_pointer, int32_t** sorted_list,
                                  int64_t* npairs) {
  return SUF(build_pbc_impl)(q, stride, N, rc, Lx, Ly, Lz, number_of_partners, key_pointer, sorted_list, npairs, 1);
}

/*
 * Count / hash mode of the same build, for boxes whose list is too large to store or beyond the reference's own
 * int32 limits (BASELINE config 4: 2.5e9 pairs > INT32_MAX, neighlist_cpu.hpp:15,29; config 5: 2 x rc overruns
 * MAX_PARTNERS*N, :37,76-78).  Same cells (GenHash :42-59, ApplyPBC :61-66), same 13 + own cell visit
 * (MakeNeighMeshId :107-132, MakePairListNaive :239-270), same accept rule (:215-223), same owner rule
 * (pair stored on min(i,j), :225-236) -- but nothing is stored except number_of_partners[] and, per slab of z cell
 * layers, the pair count and the order-independent pair-set hash of nl_oracle_hash (rows of a slab = particles whose
 * cell layer belongs to it).  Particles are first copied into cell order (values unchanged), and cells are walked by
 * OpenMP threads; sums are order independent.
 *   slab_of_layer[mz] (or NULL = one slab): slab index of every z cell layer; nslab <= 64.
 */
int SUF(nl_oracle_count)(const REAL* q, int32_t stride, int64_t N, double rc, double Lx, double Ly, double Lz,
                         int32_t* number_of_partners, int32_t nslab, const int32_t* slab_of_layer,
                         uint64_t* hash_per_slab, int64_t* pairs_per_slab, int64_t* npairs) {
  SUF(grid) g;
  if (N < 0 || N > 2147483647LL || stride < 3 || nslab < 1 || nslab > 64) return NLO_ERR_ARG;
  int rc_ = SUF(grid_init)(&g, rc, Lx, Ly, Lz);
  if (rc_) return rc_;
  const int64_t M = g.ncell;
  int32_t* cell_of = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int64_t* cell_beg = (int64_t*)calloc((size_t)(M + 2), sizeof(int64_t));
  REAL* qs = (REAL*)malloc(sizeof(REAL) * 3 * (size_t)(N > 0 ? N : 1));
  int32_t* ids = (int32_t*)malloc(sizeof(int32_t) * (size_t)(N > 0 ? N : 1));
  int ret = NLO_OK;
  if (!cell_of || !cell_beg || !qs || !ids) {
    ret = NLO_ERR_NOMEM;
    goto done;
  }
  for (int64_t i = 0; i < N; i++) { /* MakeMeshidOfPtcl :134-144 */
    const int64_t h = SUF(hash_pos)(&g, q + (size_t)i * stride);
    if (h < 0) {
      ret = NLO_ERR_OUT_OF_BOX;
      goto done;
    }
    cell_of[i] = (int32_t)h;
    cell_beg[h + 2]++;
  }
  for (int64_t c = 0; c < M; c++) cell_beg[c + 2] += cell_beg[c + 1];
  for (int64_t i = 0; i < N; i++) { /* MakeNextDest :146-165 (stable), positions carried along */
    const int64_t s = cell_beg[cell_of[i] + 1]++;
    ids[s] = (int32_t)i;
    qs[3 * s] = q[(size_t)i * stride], qs[3 * s + 1] = q[(size_t)i * stride + 1], qs[3 * s + 2] = q[(size_t)i * stride + 2];
  }
  memset(number_of_partners, 0, sizeof(int32_t) * (size_t)N);
  for (int s = 0; s < nslab; s++) hash_per_slab[s] = 0, pairs_per_slab[s] = 0;
  const int64_t layer = (int64_t)g.m[0] * g.m[1];
#pragma omp parallel
  {
    uint64_t h_loc[64];
    int64_t p_loc[64];
    for (int s = 0; s < 64; s++) h_loc[s] = 0, p_loc[s] = 0;
#pragma omp for schedule(dynamic, 16)
    for (int64_t c = 0; c < M; c++) {
      const int32_t iz = (int32_t)(c / layer), iy = (int32_t)((c % layer) / g.m[0]), ix = (int32_t)(c % g.m[0]);
      int32_t neigh[13];
      int k = 0;
      for (int32_t jz = -1; jz < 2 && k < 13; jz++) /* MakeNeighMeshId :107-132 */
        for (int32_t jy = -1; jy < 2 && k < 13; jy++)
          for (int32_t jx = -1; jx < 2 && k < 13; jx++) {
            int32_t idx[3] = {ix + jx, iy + jy, iz + jz};
            SUF(apply_pbc)(&g, idx);
            neigh[k++] = (int32_t)SUF(hash_idx)(&g, idx);
          }
      const int64_t ib = cell_beg[c], ie = cell_beg[c + 1];
      for (int64_t a = ib; a < ie; a++) {
        const int32_t pi = ids[a];
        const REAL* qi = qs + 3 * a;
        for (k = 0; k <= 13; k++) {
          int64_t jb, je;
          if (k < 13) jb = cell_beg[neigh[k]], je = cell_beg[neigh[k] + 1];
          else jb = a + 1, je = ie;
          for (int64_t b = jb; b < je; b++) {
            if (!SUF(accept)(qi, qs + 3 * b, g.rc2)) continue;
            const int32_t pj = ids[b];
            const int32_t lo = pi < pj ? pi : pj, hi = pi < pj ? pj : pi; /* :225-232 */
#pragma omp atomic
            number_of_partners[lo]++;
            const int sl = slab_of_layer ? slab_of_layer[cell_of[lo] / layer] : 0;
            uint64_t v = ((uint64_t)(uint32_t)lo << 32) | (uint32_t)hi;
            v *= 0x9E3779B97F4A7C15ULL;
            v ^= v >> 29;
            h_loc[sl] += v;
            p_loc[sl]++;
          }
        }
      }
    }
#pragma omp critical
    for (int s = 0; s < nslab; s++) hash_per_slab[s] += h_loc[s], pairs_per_slab[s] += p_loc[s];
  }
  {
    int64_t P = 0;
    for (int s = 0; s < nslab; s++) P += pairs_per_slab[s];
    *npairs = P;
  }
done:
  free(cell_of);
  free(cell_beg);
  free(qs);
  free(ids);
  return ret;
}
