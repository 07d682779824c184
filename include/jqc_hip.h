/*
 * jqc_hip.h -- C ABI of libjqc_hip.so, the MI355X (gfx950) backend of the JoltQC hot path.
 *
 * The reference has no compiled extension: its "FFI" is CuPy RawModule/RawKernel, i.e. "compile this
 * class-specialised source, give me a launchable handle, launch it on raw device pointers"
 * (/root/reference/jqc/backend/jk_1q1t.py:117-148, jk_tasks.py:40-109, linalg_helper.py:125-211).
 * Each entry point below replaces one of those generator/launcher pairs.  Plain pointers and sizes
 * only; all pointers named *_d are device pointers owned by the caller; kernels accumulate into
 * vj/vk (caller zero-fills), never allocate, free or retain them.  `stream` is a hipStream_t
 * (NULL = default stream).  Every function returns 0 on success and a negative code on failure;
 * jqc_last_error() returns the message.  Launches are asynchronous.
 */
#ifndef JQC_HIP_H
#define JQC_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JQC_BASIS_STRIDE 12 /* [x,y,z,ao_loc | c0,e0,c1,e1,c2,e2 | nprim, l]  (jqc/constants.py:33, basis.py:280-371) */
#define JQC_LMAX 4          /* jqc/constants.py:21 */
#define JQC_NPRIM_MAX 3     /* jqc/constants.py:26 */

/* algorithm ids for jqc_gen_jk_kernel (reference router: jqc/backend/jk.py:57-115) */
#define JQC_ALGO_1Q1T 0 /* one quartet per lane        (reference jk_1q1t.py / jk/1q1t.cu)  */
#define JQC_ALGO_TILE 1 /* lane-group per quartet, LDS Fock tiles (replaces jk_1qnt.py / jk/1qnt.cu) */
#define JQC_ALGO_TILE1Q 2 /* one quartet per lane inside the same LDS tile framework (small classes) */
#define JQC_ALGO_TILE512 3 /* JQC_ALGO_TILE with 512-thread workgroups: two waves per SIMD on one set of LDS tiles */
#define JQC_ALGO_PAIRVJ 8  /* pair-based J kernel (jqc_gen_pair_vj_kernel) */
#define JQC_ALGO_JKGRAD 9  /* nuclear-gradient kernel of the two-electron energy (jqc_gen_jk_grad_kernel) */
/* Tuning variant of a tiled kernel, OR-ed into the `algo` argument of jqc_gen_jk_kernel (gfx950 scheme table): */
#define JQC_VARIANT_MINW(n) ((n) << 4) /* waves per SIMD the register allocation leaves room for (0 = kernel default) */
#define JQC_VARIANT_RYS_L2 (1 << 8)    /* read the Rys table through L2 instead of staging it in LDS */
#define JQC_VARIANT_ST1 (1 << 9)       /* single-buffered TRR array; no effect while the double buffer is disabled (jk_tile.hip) */
#define JQC_VARIANT_WSYNC (1 << 10)    /* row-lane mode with every quartet inside one wave: no workgroup barrier per step */
#define JQC_VARIANT_CJR (1 << 11)      /* row-lane mode, lane = bra component i only, the j components in registers (small kets) */
#define JQC_VARIANT_NKS(log2n) ((log2n) << 12) /* lane-per-quartet mode: 2 (log2n=1) or 4 (2) ket tile pairs staged and screened
                                          per iteration, so classes with few candidates per tile pair still fill 256 lanes */

#define JQC_VARIANT_QIL (1 << 14)  /* lane-per-quartet mode: strided read of the survivor queue (neighbouring lanes take quartets
                                      of different ket slots) + four replicas of the J_ij tile: fewer same-address LDS atomics */
#define JQC_VARIANT_CORD (1 << 15) /* lane-per-quartet mode: contraction row by row with the next row's density reads issued
                                      before the current row's LDS atomics */
#define JQC_VARIANT_ECAP(code) ((code) << 16) /* row-lane mode: integrals held per lane and chunk capped at 64 (0), 32 (1), 16 (2)
                                      or 48 (3): smaller chunks re-read the TRR array from LDS but keep the lane out of scratch */

#define JQC_VARIANT_ORED (1 << 18)  /* row-lane mode: owner reduction -- the outputs that are sums over the lanes of a quartet go through
                                      a per-wave scratch area (aliasing the TRR array) and are added to the LDS Fock tiles once per
                                      quartet by one owner lane each, instead of one same-address ds_add_f64 per lane */
#define JQC_VARIANT_PAROOT (1 << 19) /* row-lane mode: a phase-A job = (quartet, root), three axes per job (Rys root evaluated once) */
#define JQC_VARIANT_NDM2 (1 << 20)  /* two density matrices contracted per integral evaluation (D / Fock tiles of both in LDS); the
                                      kernel walks n_dm in pairs.  Lane-per-quartet builds and owner-reduction builds only */
#define JQC_VARIANT_RSPLIT(code) ((code) << 22) /* row-lane mode: the Rys roots go through phase A / phase B in code + 1 groups, so the
                                      TRR array in LDS holds nroots / (code + 1) roots per quartet (room for a second workgroup per CU) */
#define JQC_VARIANT_QUAD (1 << 24)  /* lane-per-quartet mode, classes with a p shell and <= 4 Rys roots: one quartet per QUAD of lanes --
                                     * lane c of the quad owns Cartesian axis c (its recurrences, a third of the integral block), the
                                     * 1-D arrays of the neighbouring axes arrive by quad-permute DPP moves, each lane evaluates one Rys
                                     * root; 64 quartets per pass, <= 256 registers (no AGPR copies) for the 100-180-integral classes */
#define JQC_VARIANT_QCHUNK(code, index) (((code) << 25) | ((index) << 27)) /* quad builds of the 270-330-integral classes: the lane's third of
                                     * the integral block is evaluated in 2 / 3 / 5 chunks (code 1 / 2 / 3) over the Cartesian components of
                                     * shell `index` (0..3 = i, j, k, l; not the split p shell), recurrences and roots redone per chunk */
#define JQC_VARIANT_HB (1 << 29)   /* row-lane mode, "h form" (needs JQC_VARIANT_ORED, excludes JQC_VARIANT_CJR): phase A runs the transfer
                                    * AND the bra horizontal recurrence once per (quartet, root, axis) and leaves h[i][j][c] in LDS; a
                                    * phase-B lane = (bra component i, group of j components) reads one row per (axis, j component), runs
                                    * the ket recurrence and multiplies -- the bra recurrence is no longer redone per lane, root and chunk */
#define JQC_VARIANT_HEJ(code) ((code) << 25) /* h form: j components per lane = the largest divisor of nf_j that is <= 1 (code 0), 2 (1),
                                    * 3 (2), 6 (3); shares its bits with JQC_VARIANT_QCHUNK (quad builds are lane-per-quartet builds) */
#define JQC_VARIANT_KW (1 << 30)   /* JQC_ALGO_TILE512 builds with JQC_VARIANT_ORED (no JQC_VARIANT_WSYNC) of classes whose integral block
                                    * needs TWO chunks over the ket components k: the chunks run on different waves of the 512-thread
                                    * workgroup at the same time and share the quartets' recurrence arrays -- phase A once per step instead
                                    * of once per chunk, its jobs dealt over all 512 lanes, every Rys root in one pass */
#define JQC_VARIANT_MIXED (1 << 21) /* FP64 lane-per-quartet build with BOTH precision windows in one launch: quartets with an
                                      estimate above cut_hi in FP64, those in (cut_lo, cut_hi] in FP32, two per lane as packed
                                      2-vectors (v_pk_fma_f32); one staging / screening / flush per tile pair (replaces the reference's
                                      fp32 + fp64 launch pair per class, jqc/pyscf/jk.py:293-328) */

const char* jqc_last_error(void);
const char* jqc_version(void);
/* tag (hash of the kernel sources) embedded in every cached code-object name; valid after jqc_set_kernel_dirs */
const char* jqc_source_tag(void);
/* the same for the gradient kernels (jk_grad.hip on top of the sources above) */
const char* jqc_grad_source_tag(void);
/* the same for the pair-based J kernels (pair_vj.hip) */
const char* jqc_pair_source_tag(void);

/* Runtime set-up.  src_dir holds the kernel sources (joltqc_amd/csrc/kernels), cache_dir receives
 * the gfx950 code objects (one .hsaco per class/variant; replaces CuPy's cubin cache, examples/04). */
int jqc_set_kernel_dirs(const char* src_dir, const char* cache_dir);

/* Shells per tile edge of the tiled J/K kernels for l = 0..4 (default 8,4,4,2,1).  Must match the padding of the
 * shell table (BasisLayout.from_mol(alignment=tile_width)); part of the code-object cache key. */
int jqc_set_tile_widths(const int* widths5);

/* Upload the packed Rys tables (layout: joltqc_amd/backend/rys.py) to the current device. */
int jqc_set_rys_tables(const double* blob_host, size_t ndoubles);

/* gen_jk_kernel (jqc/backend/jk.py:57): returns a handle >= 0.  With compile_only != 0 the code object
 * is built into the cache (works without a GPU) and not loaded. */
int jqc_gen_jk_kernel(int li, int lj, int lk, int ll, int do_j, int do_k, int rys_lr, int fp32, int algo,
                      int compile_only);

/* Launch of a generated J/K kernel (argument order of rys_1q1t_vjk, jqc/backend/jk/1q1t.cu:45-52):
 *   basis_d  real[nbas*12], dm_d real[n_dm*nao*nao], vj_d/vk_d double[n_dm*nao*nao] (NULL if unused),
 *   quartets_d ushort4[...]; the task count is read ON DEVICE from *ntasks_d (no host sync);
 *   ntasks_max bounds the grid; qstride = +1 (list grows upward) or -1 (list grows downward). */
int jqc_jk_launch(int handle, int nao, const void* basis_d, const void* dm_d, double* vj_d, double* vk_d,
                  double omega, const void* quartets_d, const uint32_t* ntasks_d, int64_t ntasks_max,
                  int qstride, int n_dm, void* stream);

/* Launch of a tiled J/K kernel (JQC_ALGO_TILE / JQC_ALGO_TILE1Q).  No quartet queue: one workgroup per (bra tile
 * pair, chunk of consecutive ket tile pairs); screening (same predicate as jqc_screen_jk_tasks) happens inside the
 * workgroup.
 *   tasks_d     int32[ntasks][8] = {ij0, nij, kl0, nkl, nchunk, blk0, cnt, kchunk | nsplit<<16}: rectangle of the two
 *               tile-pair lists; the row owns nij*nchunk*nsplit workgroups starting at blk0, nchunk = ceil(nkl / kchunk);
 *               nsplit >= 1 workgroups share one (bra pair, ket chunk), each taking a contiguous range of candidates
 *   nblocks     total workgroups of the launch
 *   tpair_sh_d  uint32[...] = first shell of tile i <<16 | first shell of tile j; tpair_q_d = max log-Schwarz of the
 *               pair, each list sorted descending (a workgroup stops at the first ket pair below the cutoff)
 *   q_cond_d, log_dm_d  float[nbas*nbas];  processes quartets with cut_lo < q_ij+q_kl+d_large <= cut_hi
 *               (JQC_VARIANT_MIXED builds: every quartet above cut_lo -- FP64 above cut_hi, FP32 in (cut_lo, cut_hi])
 *   counter_d   optional uint64[]: the number of quartets evaluated by a workgroup is added to counter_d[cnt]
 *               (per-class dispatch counters = the "ERI quartets/s" metric, reference jk.py:288-330)
 *   counter32_d optional uint64[], JQC_VARIANT_MIXED builds: the quartets of the FP32 window go to counter32_d[cnt]
 *               (and only the FP64 ones to counter_d); ignored by every other build
 *   blk_index_d int32[nblocks/256 + 1]: task row of every 256th workgroup (coarse index of the blk0 column)
 *   tpair_ao_d  uint32[...] = first AO of tile i <<16 | first AO of tile j (same indexing as tpair_sh_d; nao < 65536)
 *   tpair_pp_d  uint32[...] = offset (units of 27 reals) of the tile pair's block in pair_tab_d, the primitive-pair
 *               prefactor table built by jqc_pair_table (real = double, or float for fp32 kernels)
 * Tile widths per angular momentum: jqc_set_tile_widths (default 8 shells for s, 4 for p and d, 2 for f, 1 for g);
 * every (l,nprim) group of the shell table must be padded to that multiple
 * (BasisLayout.from_mol(alignment=tile_width)). */
int jqc_jk_tile_launch(int handle, int nao, const void* basis_d, const void* dm_d, double* vj_d, double* vk_d,
                       double omega, const int32_t* tasks_d, int ntasks, int nblocks, const uint32_t* tpair_sh_d,
                       const float* tpair_q_d, const float* q_cond_d, const float* log_dm_d, int nbas, float cut_lo,
                       float cut_hi, float log_max_dm, int n_dm, uint64_t* counter_d, const int32_t* blk_index_d,
                       const uint32_t* tpair_ao_d, const uint32_t* tpair_pp_d, const void* pair_tab_d, uint64_t* counter32_d,
                       void* stream);

/* Screening + queue generation (replaces screen_jk_tasks, jqc/backend/jk/screen_jk_tasks.cu:75-340).
 * One launch handles a whole chunk of "screen tasks"; task t covers the rectangle
 * pairs[ij0 .. ij0+nij) x pairs[kl0 .. kl0+nkl) and appends survivors to region `cls`:
 *   tasks_d      int32[ntasks][8] = {ij0, nij, kl0, nkl, cls, blk0, 0, 0}, blk0 = first block of the task
 *   pair_sh_d    uint32[npairs]   = ish<<16 | jsh  (ish >= jsh), pair_q_d float[npairs] = log Schwarz bound
 *   log_dm_d     float[nbas*nbas] = log max|D| per shell block  (max_block_pooling + log)
 *   region_d     int64[ncls][2]   = {begin, end} of each class region in queue_d (ushort4 units)
 *   counters_d   uint32[ncls][2]  = {n_fp64 (appended from begin), n_fp32 (appended down from end)}
 * Predicate (natural-log float32, screen_jk_tasks.cu:202-261):
 *   keep if ish*nbas+jsh >= ksh*nbas+lsh and q_ij + q_kl + d_large > log_cutoff_fp32 where
 *   d_large = max(-36.8, [do_k] d_ik,d_jk,d_il,d_jl, [do_j] d_ij,d_kl);  FP64 list if > log_cutoff_fp64. */
int jqc_screen_jk_tasks(const int32_t* tasks_d, int ntasks, int nblocks, const uint32_t* pair_sh_d,
                        const float* pair_q_d, const float* log_dm_d, int nbas, int do_j, int do_k,
                        float log_cutoff_fp32, float log_cutoff_fp64, float log_max_dm, void* queue_d,
                        const int64_t* region_d, uint32_t* counters_d, void* stream);

/* max_block_pooling (jqc/backend/linalg_helper.py:125-211): out[I,J] = max_b max_{r in I, c in J} |M[b,r,c]|
 * over shell blocks given by ao_loc_d int32[nbas+1]; M is double[n_dm,nao,nao]; out float[nbas*nbas]. */
int jqc_shell_block_max(const double* mat_d, int n_dm, int nao, const int32_t* ao_loc_d, int nbas,
                        float* out_d, void* stream);

/* Primitive-pair prefactor table of the tile-pair lists (once per geometry; role of the reference's cached K_ab,
 * jqc/backend/jk/1q1t.cu:146-171).  For tile pair t (widths wi<<16|wj in tpair_wij_d) the block
 * out_d[pp_off_d[t]*27 ...] holds {c_a c_b K_ab, 1/(a+b), a+b} at ((a*wj + b)*9 + p1*3 + p2)*3. */
int jqc_pair_table(const double* basis_d, const uint32_t* tpair_sh_d, const uint32_t* tpair_wij_d,
                   const uint32_t* pp_off_d, int npairs, double* out_d, void* stream);

/* Frees the library's own device scratch (today: the per-stream partial-sum buffers of the split VV10 inner loop).  Optional: the
 * buffers are reused across calls and grow on demand; a host that unloads the library or resets the device calls this first. */
int jqc_release_scratch(void);

/* Scalar ECP integrals (replaces the kernel launches of the reference's get_ecp,
 * /root/reference/jqc/backend/ecp.py:1371-1503 -> ecp/ecp_type1.cu, ecp_type2.cu): for every task {ish, jsh, k} (ish <= jsh, ECP atom k)
 *   mat[ao_i.., ao_j..] += <i| U_L |j> + sum_l <i| U_l P_l |j>   (and the transposed block when ish != jsh), internal Cartesian AOs.
 * basis_d: packed shell rows (12 doubles, as every kernel); ecp_xyz_d [natm_ecp][3]; terms of ECP atom k =
 * ecp_terms_d[ecp_loc_d[k] .. ecp_loc_d[k + 1]) x {l (-1 = local channel), radial power n, zeta, coefficient}, i.e. the rows of
 * mol._ecpbas flattened per primitive; rgrid_d / wgrid_d: nr radial quadrature points (r, dr weight, WITHOUT r^2) on (0, inf);
 * ylm_d [25][15]: Cartesian monomial coefficients (libcint order) of the orthonormal real spherical harmonics l <= 4.
 * symmetric = 1: tasks with ish <= jsh, both triangles written (the value integrals); 0: only the (ish, jsh) block of every task --
 * how the first-derivative integrals <grad a|U|b> (reference get_ecp_ip, backend/ecp.py:953-1138) are assembled from l + 1 / l - 1
 * auxiliary bra shells appended to basis_d (joltqc_amd/backend/ecp.py).  lmax_shell = largest l in basis_d: up to 5 the kernel instance
 * with three radial points per chunk runs, 6 (l + 2 shells of the second derivatives, get_ecp_ipip) the one with two.  Projectors up to
 * l = 4, FP64 only (the reference's ECP kernels are FP64 only too, jqc/pyscf/ecp.py:51-53). */
int jqc_ecp_scalar(const double* basis_d, int nao, const int32_t* tasks_d, int ntasks, const double* ecp_xyz_d, const int32_t* ecp_loc_d,
                   const double* ecp_terms_d, const double* rgrid_d, const double* wgrid_d, int nr, const double* ylm_d, double* mat_d,
                   int symmetric, int lmax_shell, void* stream);

/* One-electron integrals (overlap S, kinetic T, nuclear attraction V) of n shell pairs (ish << 16 | jsh, ish >= jsh) in the
 * internal Cartesian basis, [nao, nao] each, both triangles written.  atoms_d = [x, y, z, Z] per nucleus (Bohr).  The
 * reference takes these from PySCF/libcint on the CPU (mf.get_hcore / get_ovlp); SURVEY.md 8f row 1. */
int jqc_int1e(const double* basis_d, const int32_t* ao_loc_d, const uint32_t* pairs_d, int npairs, const double* atoms_d,
              int natm, int nao, double* S_d, double* T_d, double* V_d, void* stream);

/* ---- pair-based Coulomb path (second J algorithm: jqc/backend/jk_pair.py:288-371 gen_vj_kernel, jk/pair_vj.cu:43-465) ----
 * A lane owns one bra shell pair and walks the Schwarz-sorted ket pair list; J_ij stays in registers.  The density is folded
 * into the ket side once per call by jqc_pair_ket_density (E coefficients over the combined ket index, see pair_vj.hip).
 *
 * jqc_gen_pair_vj_kernel: handle of the (li lj | lk ll) kernel, li >= lj, lk >= ll (no li >= lk restriction: bra and ket
 * classes are independent here).  Returns -4 (and builds nothing loadable) when the class does not fit the register file
 * without scratch: the caller keeps such classes on the tiled kernels. */
int jqc_gen_pair_vj_kernel(int li, int lj, int lk, int ll, int rys_lr, int compile_only);
/* number of E coefficients per ket pair of angular momenta (lk, ll): triples (cx,cy,cz), lk <= cx+cy+cz <= lk+ll */
int jqc_pair_ntrip(int lk, int ll);
/* E_d[pair][ntrip] and ld_d[pair] = log max|D_kl| for n_ket pairs (ksh << 16 | lsh) of one angular class; dm_d is the
 * SYMMETRISED density [nao, nao] in the internal Cartesian order. */
int jqc_pair_ket_density(const double* basis_d, const double* dm_d, int nao, const uint32_t* ket_pairs_d, int n_ket, int lk,
                         int ll, double* E_d, float* ld_d, void* stream);
/* vj_d[(j0+j)*nao + i0+i] += raw J_ij of every bra pair (same raw convention as the tiled kernels: the epilogue doubles and
 * adds the transpose).  Bra list: ONE (l, nprim) group pair (npi, npj uniform), sorted by bound; ket list: ket_seg_d[2 s],
 * [2 s + 1] = first entry / length of sorted segment s inside ket_pairs_d / ket_q_d / ket_ld_d / ket_tab_d / E_d.
 * *_tab_d: 27 doubles per pair (jqc_pair_table with 1 x 1 tiles).  nsplit workgroups share a block of 256 bra pairs. */
int jqc_pair_vj_launch(int handle, int nao, const double* basis_d, const double* E_d, double* vj_d, double omega,
                       const uint32_t* bra_pairs_d, int n_bra, const float* bra_q_d, const double* bra_tab_d,
                       const uint32_t* ket_pairs_d, const float* ket_q_d, const float* ket_ld_d, const double* ket_tab_d,
                       const int32_t* ket_seg_d, int nseg, float log_cut, float log_max_dm, int npi, int npj, int nsplit,
                       uint64_t* counter_d, void* stream);

/* ------------------------------------------------------------------------------------------ nuclear gradient of E2
 * SURVEY.md 8(f) row 3.  The reference has no gradient kernels (its scanners take GPU4PySCF's CUDA gradients,
 * jqc/pyscf/__init__.py:63-97); the host-side counterpart these two calls replace is GPU4PySCF's per-atom J/K energy
 * derivative (gpu4pyscf.grad.rhf `_jk_energy_per_atom(mol, dm, vhfopt, j_factor, k_factor)`).
 * jqc_gen_jk_grad_kernel: code object of csrc/kernels/jk_grad.hip for class (li lj|lk ll), canonical order; FP64 only.  The
 *   generator picks one of two forms per class (measured table): T = nf_i nf_j lanes per quartet with the 1-D
 *   derivative records in LDS, or one quartet per lane; jqc_jk_grad_launch sizes the grid for the form the handle was built in.
 * jqc_jk_grad_launch: for every quartet of the queue (same ushort4 entries and device-side count as jqc_jk_launch)
 *   grad_d[rep][atom][3] += d/dR_atom of  sum_abcd (ab|cd) [4 j_factor D_ab D_cd - k_factor n_dm sum_s (D^s_ac D^s_bd + D^s_ad D^s_bc)]
 *   with D = sum_s D^s, dm_d = n_dm (1 or 2) symmetric matrices [nao, nao] in the internal Cartesian order, shell_atom_d[nbas]
 *   = atom of every internal shell, rep = workgroup index mod nrep (the caller sums the nrep replicas).  With j_factor =
 *   k_factor = 1 and the total closed-shell density this is the gradient of 1/2 tr(D J) - 1/4 tr(D K) at fixed D. */
int jqc_gen_jk_grad_kernel(int li, int lj, int lk, int ll, int rys_lr, int compile_only);
int jqc_jk_grad_launch(int handle, int nao, const double* basis_d, const double* dm_d, int n_dm, double* grad_d,
                       const int32_t* shell_atom_d, int natm, int nrep, double j_factor, double k_factor, double omega,
                       const void* quartets_d, const uint32_t* ntasks_d, int64_t ntasks_max, int qstride, void* stream);

/* Schwarz bounds on device (replaces the libcvhf call in compute_q_matrix, jqc/pyscf/basis.py:840-867):
 * for each listed pair p = (ish<<16|jsh) with l(ish)=li, l(jsh)=lj:  out[p] = sqrt(max_ab |(ab|ab)|). */
int jqc_schwarz(int li, int lj, const double* basis_d, const uint32_t* pair_sh_d, int npairs, double omega,
                double* out_d, void* stream);

/* ------------------------------------------------------------------------------------------ DFT grid path
 * Grid coordinates are SoA double[3][ngrids], ngrids a multiple of 256 (reference: jqc/backend/rks.py:84-86),
 * sorted so that each block of 256 points is spatially compact (jqc/pyscf/rks.py:148-168).
 *
 * jqc_dft_ao_screen  replaces estimate_log_aovalue (jqc/backend/rks.py:167-242, dft/estimate_log_aovalue.cu:36):
 *   per block the shells whose log-max AO estimate exceeds log_cutoff, ascending, zero-width (padding) shells excluded:
 *   shell_list uint16[nblk*nbas], row_of int32[nblk*nbas] (first AO row of the shell inside the block),
 *   nshl int32[nblk], nrow int32[nblk].
 * jqc_dft_eval_ao    AO values (ncomp=1) or values+gradient (ncomp=4) of blocks [blk0, blk0+nblk) into the workspace
 *   ws double[ncomp][rows][256] (comp_stride = rows*256 doubles), block b starts at row row_base[b] (multiple of 16,
 *   block rows padded to a multiple of 16 with zeros); ao_idx int32[rows] = internal AO index of a row or -1.
 * jqc_dft_rho        replaces eval_rho (jqc/backend/rks.py:40-96, dft/eval_rho.cu:58): rho[ndim][ngrids] +=, ndim 1/4/5,
 *   dm double[nao*nao] symmetric, internal Cartesian order.
 * jqc_dft_vxc        replaces eval_vxc (jqc/backend/rks.py:104-160, dft/eval_vxc.cu:87): vmat[nao*nao] += the FULL
 *   symmetric V (the reference accumulates the upper triangle and symmetrises afterwards); wv[ndim][ngrids] already
 *   contains the quadrature weights.
 * jqc_vv10           replaces vv10_kernel (jqc/backend/rks.py:250-335, dft/vv10.cu:29): F,U,W double[ngrids].
 *   fp32: 0 = FP64 inner loop, 1 = FP32 inner loop (the reference's), 3 = FP32 and the caller guarantees K, Kp >= 1e-3 at
 *   every point (true for VV10's K = b k_F-type factors of points with rho >= 1e-10, padding included): the reference's
 *   test `g' (g gt)^2 > 1e-30` (vv10.cu:104) can then not fail and is not evaluated. */
int jqc_dft_ao_screen(const double* coords_d, int ngrids, const double* basis_d, const int32_t* ao_loc_d, int nbas,
                      float log_cutoff, uint16_t* shell_list_d, int32_t* row_of_d, int32_t* nshl_d, int32_t* nrow_d,
                      float* shell_la_d, void* stream);
int jqc_dft_eval_ao(const double* coords_d, int ngrids, const double* basis_d, int nbas, int blk0, int nblk,
                    const uint16_t* shell_list_d, const int32_t* row_of_d, const int32_t* nshl_d, const int32_t* nrow_d,
                    const int64_t* row_base_d, int ncomp, int64_t comp_stride, double* ws_d, int32_t* ao_idx_d,
                    const float* shell_la_d, float* row_la_d, void* stream);
/* row_la_d: log estimate of every workspace row (sorted, largest first, per block); AO pairs (a, b) of a block with
 * la_a + la_b above thr64 are contracted in FP64, between thr32 and thr64 in FP32 (MFMA f32), below thr32 not at all
 * (the reference's [cutoff_a, cutoff_b) windows, eval_rho.cu:93-106; thr = log cutoff - log max|D| resp. log max|wv|).
 * order_d (may be NULL): permutation of 0..nblk-1, workgroup b of the launch takes block blk0 + order_d[b] -- the host passes the
 * blocks by descending AO-row count so that the longest ones do not start last. */
int jqc_dft_rho(int blk0, int nblk, int ngrids, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                const double* ws_d, const int32_t* ao_idx_d, const double* dm_d, int nao, int ndim, double* rho_d,
                const float* row_la_d, float thr64, float thr32, const int32_t* order_d, void* stream);
int jqc_dft_vxc(int blk0, int nblk, int ngrids, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                const double* ws_d, const int32_t* ao_idx_d, const double* wv_d, int ndim, int nao, double* vmat_d,
                const float* row_la_d, float thr64, float thr32, const int32_t* order_d, void* stream);
/* Nuclear gradient of E_xc at fixed density and fixed grid (LDA: ndim 1, GGA: ndim 4, meta-GGA: ndim 5; SURVEY.md 8(f) row 3 -- no reference
 * kernel: JoltQC leaves gradients to GPU4PySCF, whose `get_vxc`-type gradient drivers these two calls serve).
 * jqc_dft_xcgrad_ao: like jqc_dft_eval_ao with EIGHT workspace components per AO row, [phi, X, d_x phi, d_y phi, d_z phi, H^x,
 *   H^y, H^z], X = wv0 phi + sum_y wv_y d_y phi, H^x = sum_y wv_y d_x d_y phi, wv_d[ndim][ngrids] = weights x vxc; ndim = 5
 *   (meta-GGA): FOURTEEN components, the six added ones T_uv = 1/2 wv4 d_u d_v phi (xx, xy, xz, yy, yz, zz).
 * jqc_dft_xcgrad: gao_d[ao][3] += -2 sum_g (d_x phi_a (D X)_a + H^x_a (D phi)_a) for the AO rows of blocks [blk0, blk0+nblk);
 *   AO pairs with la_a + la_b <= thr are skipped; FP64 MFMA.  The caller sums AOs into atoms. */
int jqc_dft_xcgrad_ao(const double* coords_d, int ngrids, const double* basis_d, int nbas, int blk0, int nblk,
                      const uint16_t* shell_list_d, const int32_t* row_of_d, const int32_t* nshl_d, const int32_t* nrow_d,
                      const int64_t* row_base_d, const double* wv_d, int ndim, int64_t comp_stride, double* ws_d,
                      int32_t* ao_idx_d, const float* shell_la_d, float* row_la_d, void* stream);
int jqc_dft_xcgrad(int blk0, int nblk, const int32_t* nrow_d, const int64_t* row_base_d, int64_t comp_stride,
                   const double* ws_d, const int32_t* ao_idx_d, const double* dm_d, int nao, double* gao_d,
                   const float* row_la_d, float thr, const int32_t* order_d, int ndim, void* stream);
int jqc_vv10(double* F_d, double* U_d, double* W_d, const double* vvcoords_d, const double* coords_d,
             const double* W0p_d, const double* W0_d, const double* K_d, const double* Kp_d, const double* RpW_d,
             int vvngrids, int ngrids, int fp32, void* stream);

#ifdef __cplusplus
}
#endif
#endif
