/*
 * astrild_hip.h — C-ABI of libastrild_hip.so, the MI355X (gfx950) drop-in for
 * the numerics underneath astrild's post-processing hot path.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes; no exceptions cross the boundary.
 *   - Every pointer named *_d / documented "device" is HBM (hipMalloc'd, or
 *     the data_ptr() of a PyTorch-ROCm tensor used as a memory holder).
 *   - The caller owns every buffer.  Plans own only their rocFFT work area.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); the
 *     call enqueues work on it and returns without synchronising, unless the
 *     doc says "synchronous".
 *   - Return value: AST_OK (0) or a negative AST_ERR_*; ast_last_error()
 *     returns a thread-local message for the last failure.
 *   - dtype: AST_F32 / AST_F64 selects the element type of particle and grid
 *     buffers (spectra are the matching interleaved complex type).
 *
 * Each entry cites the reference interface it replaces (paths relative to
 * /root/reference/src/astrild/).
 */
#ifndef ASTRILD_HIP_H
#define ASTRILD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AST_OK 0
#define AST_ERR_ARG (-1)
#define AST_ERR_HIP (-2)
#define AST_ERR_ROCFFT (-3)
#define AST_ERR_WORKSPACE (-4)

#define AST_F32 0
#define AST_F64 1

#define AST_WIN_NGP 0
#define AST_WIN_CIC 1
#define AST_WIN_TSC 2

#define AST_FFT_R2C 0 /* real forward  */
#define AST_FFT_C2R 1 /* real inverse  */
#define AST_FFT_C2C_FWD 2
#define AST_FFT_C2C_INV 3

int ast_version(void);
const char* ast_last_error(void);

/* Per-launch HIP-event timing for bench.py's roofline leg: while enabled every
 * kernel launch site records an event pair on its stream.  ast_profile_report
 * synchronises them and writes one "name,calls,total_ms" line per site into
 * buf (NUL terminated).  ast_profile_enable(0/1) also clears the records. */
int ast_profile_enable(int on);
int ast_profile_report(char* buf, size_t cap);

/* ---------------------------------------------------------------- utility */

/* buf[i] = value, i < count. */
int ast_fill(void* buf_d, int dtype, size_t count, double value, void* stream);

/* buf[i] /= divisor (IEEE division, bit-exact with numpy).  Replaces
 * SkyUtils.convert_code_to_phy_units (rays/skys/sky_utils.py:318-339:
 * kappa/shear/deflt / c^2, isw_rs / c^3) and the `/ dx**3` of
 * particles/hutils/stats_subfind.py:132. */
int ast_divide(void* buf_d, int dtype, size_t count, double divisor, void* stream);

/* Synthetic "Gaussian-random" particle set of SURVEY.md §8(d), generated in
 * HBM: lattice point q = (i+1/2, j+1/2, k+1/2) L/npside (k fastest) plus
 * sigma * N(0,1) per component, wrapped into [0, L).  Counter-based
 * generator keyed by (seed, particle index, component); particles
 * [first, first+count) of the lattice are written to pos_d as (count, 3).
 * `shuffle_stride` = UINT64_MAX writes them in a pseudo-random order, a fixed
 * permutation of the range keyed by seed + 1 (cycle-walked Feistel network over
 * the splitmix64 finaliser): the scatter-unfriendly ordering of SURVEY.md S8(d).
 * Any other non-zero value: t -> first + (t * shuffle_stride) mod count (a
 * permutation when the stride is coprime with count; low-discrepancy, kept for A/B).
 * Bench / test plumbing, not part of the reference. */
int ast_synth_lattice_particles(void* pos_d, int dtype, size_t first, size_t count,
                                int npside, double boxsize, double sigma,
                                uint64_t seed, uint64_t shuffle_stride, void* stream);
/* A CLUSTERED synthetic set (bench / test plumbing): the npside^3 lattice collapsing onto `nattractors` (<= 1024) centres,
 * x = q - sum_h amplitude exp(-r^2 / 2 R_h^2) (q - c_h) + sigma N(0,1), R_h between 6 and 48 lattice spacings (many small,
 * few large), centres and radii from hashes of seed.  amplitude 0.95: knots of several 10^5 particles in a few cells -
 * tile occupancies ~100 x the mean, like the evolved snapshots and halo catalogues the reference paints
 * (stats_subfind.py:125-131) - in lattice order (neighbours in memory stay neighbours in space outside the knots) or,
 * with shuffle != 0, in the pseudo-random order of ast_synth_lattice_particles.  count must be npside^3. */
int ast_synth_clustered_particles(void* pos_d, int dtype, size_t count, int npside, double boxsize, double sigma,
                                  uint64_t seed, int nattractors, double amplitude, int shuffle, void* stream);

/* ----------------------------------------------- a-1 / a-2: mass assignment */

/* NGP scatter-ASSIGN with numpy's last-write-wins semantics.  Replaces
 * PowerSpectrum3D._read_data, power_spectra/power_spectrum_3d.py:142-148:
 *     idx = (npar * coord).astype(int);  value_map[(x, y, z)] = values
 * x_d, y_d, z_d, values_d: np elements each (DataFrame columns, SoA).
 * grid_d: npar^3 elements, C order (axis 0 slowest); zero-filled by the call.
 * owner_d: npar^3 uint32 scratch.  np must be < 2^32 - 1.  Coordinates must
 * lie in [0, 1) like the reference requires (numpy would raise IndexError
 * otherwise); out-of-range particles are counted in *dropped_d (device
 * uint64, may be NULL) and skipped. */
int ast_ngp_assign(const void* x_d, const void* y_d, const void* z_d, const void* values_d,
                   int dtype, size_t np, int npar, void* grid_d, uint32_t* owner_d,
                   unsigned long long* dropped_d, void* stream);

/* Mass-weighted scatter-ADD with a separable NGP / CIC / TSC window.
 * Replaces pmesh ParticleMesh(Nmesh=[n]*3, BoxSize=L).paint(pos, mass=,
 * resampler=) as called at particles/hutils/stats_subfind.py:130-131 (and the
 * window='TSC' meshes of power_spectrum_3d.py:197-212).
 *   pos_d:  (np, 3) AoS, box units; any real value wraps periodically.
 *   mass_d: np elements or NULL (unit mass).
 *   scale:  every deposit is multiplied by it (1/dx^3 folds the division of
 *           stats_subfind.py:132 into the paint; 1.0 = pmesh behaviour).
 *   grid_d: ACCUMULATED into (caller zero-fills).  It holds nx_alloc planes
 *           of nmesh x nmesh cells: buffer plane p is global axis-0 plane
 *           (x_start + p) mod nmesh.  Single GPU: x_start = 0,
 *           nx_alloc = nmesh.  Slab-decomposed: the rank's owned planes plus
 *           its ghost planes.  Deposits falling outside the buffer are
 *           counted in *dropped_d (device uint64, may be NULL) and skipped.
 *   shift_cells: added to every coordinate in GRID units, s = x * nmesh/boxsize + shift_cells (one fma;
 *           0 = plain paint).  Interlacing (nbodykit CatalogMesh, interlaced=True) paints a second
 *           mesh with shift_cells = 0.5, see ast_interlace_compensate.
 * Index/fraction arithmetic is float64 for both dtypes; the accumulation is
 * in `dtype` with hardware atomics (sum order is not reproducible). */
int ast_paint(int window, int dtype, const void* pos_d, const void* mass_d, size_t np,
              int nmesh, double boxsize, double scale, int x_start, int nx_alloc,
              void* grid_d, unsigned long long* dropped_d, double shift_cells, void* stream);

/* Same result as ast_paint, for the CIC/TSC windows, through LDS-resident
 * grid tiles: particles are run-length grouped by the tile of their base
 * cell (no particle data is moved), each workgroup accumulates one tile in
 * LDS and flushes it once.  workspace_d must hold
 * ast_paint_tiled_workspace_bytes(window, dtype, np, nmesh, nx_alloc, flags) bytes. */
/* flags: 0 = single pass over the particles (fixed-capacity tile segments, overflow
 * deposited with global atomics); AST_PAINT_TWO_PASS = exact count + scan first
 * (one more read of the positions, no overflow path: for strongly clustered input). */
#define AST_PAINT_TWO_PASS 1
/* AST_PAINT_OVERWRITE: grid_d is OVERWRITTEN with this paint instead of accumulated into
 * (no zero-fill needed).  Each workgroup then walks a whole z-column of tiles with the z
 * halo carried in LDS; owned cells leave as plain stores, the x/y halo ring goes through
 * per-column records folded in by a second kernel — no global float atomics in the flush.
 * The LDS tiles then accumulate in 64-bit fixed point (order-independent, so the paint is
 * bit-reproducible); `mass_bound` must be >= max |mass| when mass_d is given (ast_minmax). */
#define AST_PAINT_OVERWRITE 2
/* AST_PAINT_DEFER_FOLD (with AST_PAINT_OVERWRITE, whole periodic grid only): the last kernel of the
 * paint - adding the columns' x/y halo records into their neighbours' border lines - is left out;
 * the grid is complete only after ast_fft_tile_power_3d_halo has read it, which folds the records
 * while its z pass loads the rows (one kernel and ~2 GB of traffic less at 1024^3). */
#define AST_PAINT_DEFER_FOLD 4
/* AST_PAINT_SCATTERED (single pass + AST_PAINT_OVERWRITE): a hint that the particles have no spatial order
 * in memory.  The single-pass overwrite paint groups each 32-particle window by tile (8-byte group records)
 * and copies the few particles that have no companions in their window ("strays", 16/32 bytes) into their
 * tile's segment; by default a tile's stray segment holds a quarter of its particle capacity, with this flag
 * all of it (3.4x the workspace).  Without the hint unordered input still paints correctly - what does not
 * fit goes through the (slow) overflow list. */
#define AST_PAINT_SCATTERED 8
/* AST_PAINT_XSORTED (single pass + AST_PAINT_OVERWRITE, not with AST_PAINT_SCATTERED): a hint that the particles
 * come in ascending x (buffer planes) - lattice order, slab-ordered snapshot files.  The paint then runs as a
 * pipeline over chunks of particles: while chunk k + 1 is grouped, the column walk of the tile rows that chunk
 * k completed runs beside it on a second stream (event-ordered against `stream`).  A particle that arrives for a
 * row already walked goes through the overflow list (global atomics): input that is not sorted after all still
 * paints correctly, only slower. */
#define AST_PAINT_XSORTED 16
/* offset (AST_PAINT_OVERWRITE only, else 0): every OWNED cell is stored as (sum - offset), the
 * subtraction done in double on the exact fixed-point sum before the single rounding to `dtype`
 * (halo records stay additive).  With offset = total mass * scale / nmesh^3 the grid holds the
 * density CONTRAST rho - mean: only the DC mode of its spectrum changes (FFTPower discards it),
 * and fp32 cells stop carrying the rounding error of the O(1) mean.  Only the buffer planes
 * [offset_start, offset_start + offset_count) receive it (offset_count < 0: all): the ghost planes of a slab
 * buffer are ADDED to other ranks' cells and must stay plain sums. */
size_t ast_paint_tiled_workspace_bytes(int window, int dtype, size_t np, int nmesh, int nx_alloc, int flags);
int ast_paint_tiled(int window, int dtype, const void* pos_d, const void* mass_d, size_t np,
                    int nmesh, double boxsize, double scale, int x_start, int nx_alloc,
                    void* grid_d, void* workspace_d, size_t workspace_bytes,
                    unsigned long long* dropped_d, int flags, double mass_bound, double offset,
                    int offset_start, int offset_count, double shift_cells, void* stream);
/* The single-pass AST_PAINT_OVERWRITE paint in parts (no TWO_PASS / DEFER_FOLD / XSORTED), for callers that consume the
 * grid plane range by plane range while the rest is still being painted - the slab pipeline (the
 * reference paints on one rank, stats_subfind.py:130-131): every call takes the arguments of ast_paint_tiled, all of
 * them identical from call to call, plus
 *   AST_PAINT_STAGE_GROUP          once: the particle lists of every tile (and the scatter levels of AST_PAINT_SCATTERED);
 *   AST_PAINT_STAGE_WALK  row0 n   the column walk (and z seams) of tile rows [row0, row0 + n): their owned cells are
 *                                  stored, their x / y halo goes to the rows' records;
 *   AST_PAINT_STAGE_FOLD  row0 n   adds the neighbours' records (and the overflow list's deposits) into these rows'
 *                                  planes: needs the WALK of rows row0 - 1 .. row0 + n (CIC: row0 - 1 .. row0 + n - 1)
 *                                  that exist in the buffer, and completes buffer planes [row0 * P, (row0 + n) * P),
 *                                  P = ast_paint_tile_row_planes().
 * A tile row is P consecutive buffer planes, row 0 starting at buffer plane 0; ast_paint_tile_rows(nx_alloc) rows.
 * After GROUP, WALK of every row and FOLD of every row the grid equals ast_paint_tiled's, bit for bit.  Calls may go to
 * different streams; ordering them as above is the caller's business. */
#define AST_PAINT_STAGE_ALL (-1)
#define AST_PAINT_STAGE_GROUP 0
#define AST_PAINT_STAGE_WALK 1
#define AST_PAINT_STAGE_FOLD 2
/* Grouping in PARTS, for particles that come in ascending x (slab-ordered input; slab buffers only, not AST_PAINT_SCATTERED):
 *   AST_PAINT_STAGE_RESET          once, instead of GROUP: clears the lists' counters;
 *   AST_PAINT_STAGE_GROUP_PART k K the lists of part k of K equal parts of the particle array (row0 = k, nrows = K), in any
 *                                  order of parts; nrows = K | (span - 1) << 16 (K <= 65535) takes the `span` consecutive
 *                                  parts k .. k + span - 1 in ONE launch (stages of unequal size).  closed_row0 / closed_nrows name the tile rows that have been WALKED
 *                                  already - one range, taken modulo the buffer's rows (the rows of the top ghost planes
 *                                  first, then 0, 1, ...).  A particle of a later part whose tile lies in a closed row
 *                                  cannot be painted any more (the row may have been transformed and sent): it is counted
 *                                  in *dropped_d - the caller's promise about the order did not hold - and skipped.
 * A row may be walked once every part that can hold a particle of it has been grouped; the walks, folds and the result
 * are those of the one-part GROUP.  closed_row0 / closed_nrows are ignored by the other stages. */
#define AST_PAINT_STAGE_GROUP_PART 3
#define AST_PAINT_STAGE_RESET 4
/*   AST_PAINT_STAGE_LATE  row0 n   FOLD without the record fold: only the overflow / late list's deposits into these rows'
 *                                  planes - for rows whose halo records are added by the consumer instead
 *                                  (ast_fft_tile_rows_r2c_slab_halo folds them as its z pass loads the rows). */
#define AST_PAINT_STAGE_LATE 5
int ast_paint_tiled_stage(int window, int dtype, const void* pos_d, const void* mass_d, size_t np,
                          int nmesh, double boxsize, double scale, int x_start, int nx_alloc,
                          void* grid_d, void* workspace_d, size_t workspace_bytes,
                          unsigned long long* dropped_d, int flags, double mass_bound, double offset,
                          int offset_start, int offset_count, double shift_cells, int stage, int row0, int nrows,
                          int closed_row0, int closed_nrows, void* stream);
int ast_paint_tile_rows(int nx_alloc);
int ast_paint_tile_row_planes(void);
/* Where a paint with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD and these parameters left its halo
 * records inside workspace_d (for ast_fft_tile_power_3d_halo). */
int ast_paint_tiled_halo(void* workspace_d, int window, int dtype, size_t np, int nmesh, int nx_alloc, int flags,
                         void** rec_out);

/* What the single-pass AST_PAINT_OVERWRITE paint with these parameters left in workspace_d: out_d[0] group records,
 * out_d[1] stray copies, out_d[2] particles that went through the overflow list, out_d[3] the largest number of
 * strays a tile asked for (device uint64 x 4).  Diagnostics: a large out_d[2] says the input wants
 * AST_PAINT_SCATTERED or AST_PAINT_TWO_PASS. */
int ast_paint_tiled_list_stats(void* workspace_d, int window, int dtype, size_t np, int nmesh, int nx_alloc, int flags,
                               unsigned long long* out_d, void* stream);
/* Does the input have spatial order in memory?  `windows` (2 .. 65536) runs of 32 consecutive particles, evenly spread over
 * pos_d; *groupable_d (device, 4 bytes; zeroed by the call) = the number of runs in which at least 8 particles share the
 * 8 x 8 x 32-cell tile of the run's 16th particle - what the tiled paint's grouping kernel turns into group records.
 * Lattice-, cell- or curve-ordered input: nearly all runs; shuffled input: none, and the paint belongs on
 * AST_PAINT_SCATTERED from the start (device.paint decides so below a quarter).  np >= 32.  No reference counterpart:
 * pmesh's paint (stats_subfind.py:130-131) does not care about order. */
int ast_paint_order_probe(const void* pos_d, int dtype, size_t np, int nmesh, double boxsize, double shift_cells,
                          int windows, unsigned* groupable_d, void* stream);
/* Will the single-pass paint's fixed tile segments (room for max(2 mean, mean + 512) particles per 8 x 8 x 32-cell tile)
 * overflow?  `samples` particles, one from every stride of np / samples at a hashed offset, are counted per tile of the
 * periodic grid in counts_d (ast_paint_occupancy_probe_bytes(nmesh) bytes, zeroed by the call); out_d[0] = the estimated
 * number of particles beyond their tile's capacity, out_d[1] = the largest estimated tile occupancy (device uint64 x 2).
 * With >= 8 samples per tile on average a uniform input reads < 0.1 % of np; clustered input (evolved snapshots, the halo
 * catalogue of stats_subfind.py:125-131) reads what the overflow list would have held - before anything is painted, so
 * the caller can go to AST_PAINT_TWO_PASS at once.  nmesh a multiple of 32. */
size_t ast_paint_occupancy_probe_bytes(int nmesh);
int ast_paint_occupancy_probe(const void* pos_d, int dtype, size_t np, int nmesh, double boxsize, double shift_cells,
                              size_t samples, void* counts_d, size_t counts_bytes, unsigned long long* out_d, void* stream);

/* Interlacing and window compensation of a catalogue-painted mesh in Fourier space - what nbodykit's
 * CatalogMesh does for the parameters astrild writes at power_spectra/power_spectrum_3d.py:197-212
 * (compensated=True, interlaced=True, window='TSC'; they are inert for the ArrayMesh inputs the reference
 * passes, real for particle catalogues).  c1_d: half spectrum (block [i0, i1, nmesh/2+1], like
 * ast_power_bin_1d) of the plain paint, updated in place; c2_d: spectrum of the paint with shift_cells = 0.5,
 * or NULL (no interlacing):
 *   c1 <- (c1 + c2 exp(i (wx+wy+wz)/2)) / 2,   w = 2 pi m / nmesh;
 *   compensate != 0:  c1 <- c1 / prod_axes W(w):  sinc(w/2)^p (p = 2 CIC, 3 TSC) with interlacing,
 *   sqrt(1 - 2/3 s) / sqrt(1 - s + 2/15 s^2), s = sin^2(w/2), without (nbodykit's Compensate*Shotnoise). */
int ast_interlace_compensate(void* c1_d, const void* c2_d, int dtype, int nmesh, int window, int compensate,
                             int i0_start, int i0_count, int i1_start, int i1_count, void* stream);

/* Particle routing for input that is not partitioned by slab: the destination of a particle is
 * the slab (nmesh / nparts planes of axis 0) that owns its base plane - floor(s) for CIC, floor(s + 1/2) for NGP/TSC,
 * s = x nmesh / boxsize, wrapped.  ast_route_count: counts_d[part] (device uint64, caller zero-fills) += particles per
 * slab.  ast_route_scatter: cursor_d holds the exclusive prefix sums of the counts on entry (it is advanced); slab
 * `part`'s particles are written contiguously to out_pos_d (and out_mass_d when mass_d is given) from its cursor on,
 * ready for an all-to-all-v with the counts as split sizes.  nparts <= 64. */
int ast_route_count(const void* pos_d, int dtype, size_t np, int nmesh, double boxsize, int window, int nparts,
                    unsigned long long* counts_d, void* stream);
int ast_route_scatter(const void* pos_d, const void* mass_d, int dtype, size_t np, int nmesh, double boxsize, int window,
                      int nparts, unsigned long long* cursor_d, void* out_pos_d, void* out_mass_d, void* stream);

/* dst[i] += src[i] — ghost-plane fold after a slab paint. */
int ast_accumulate(void* dst_d, const void* src_d, int dtype, size_t count, void* stream);
/* Measurement aid (SURVEY.md §8d "also measure an on-box streaming-copy kernel"): streams `bytes` (a multiple of 16,
 * 16-byte aligned) with 16-byte accesses - mode 0 copy src -> dst, 1 read src only, 2 write dst only (the fastest access
 * variant of each; mode | 256 | variant << 4 picks one explicitly, scripts/micro/copy_rate.py).  bench.py times it
 * and quotes every roofline fraction against this ceiling beside the 8 TB/s of the data sheet.  No reference counterpart. */
int ast_stream_copy(void* dst_d, const void* src_d, size_t bytes, int mode, void* stream);

/* ------------------------------------------------------------- a-4: FFTs */

typedef struct ast_fft_plan ast_fft_plan;

/* rocFFT plan.  lengths[] are in C (row-major) order, slowest axis first,
 * like numpy shapes.  Real transforms keep the half spectrum on the LAST
 * axis (lengths[rank-1]/2+1 complex values).  `scale` multiplies every
 * output element (1/Ng gives pmesh's r2c normalisation used by nbodykit
 * FFTPower, power_spectrum_3d.py:189-195).  Contiguous layouts.  The plan
 * allocates its rocFFT work buffer with hipMalloc. */
int ast_fft_plan_create(ast_fft_plan** plan, int kind, int dtype, int rank,
                        const size_t* lengths, size_t batch, double scale, int inplace);

/* Complex 1-D transforms of length `length` with element stride `stride`,
 * `batch` of them `dist` elements apart (in place) — the axis-0 pass of the
 * slab-decomposed 3D FFT. */
int ast_fft_plan_create_strided_1d(ast_fft_plan** plan, int kind, int dtype, size_t length,
                                   size_t stride, size_t batch, size_t dist, double scale);

/* General strided layout: strides (in elements of the respective array, real or
 * complex) are given per axis in the same C order as lengths[]. */
int ast_fft_plan_create_general(ast_fft_plan** plan, int kind, int dtype, int rank, const size_t* lengths,
                                const size_t* in_strides, const size_t* out_strides, size_t batch,
                                size_t in_dist, size_t out_dist, double scale, int inplace);

size_t ast_fft_plan_work_bytes(const ast_fft_plan* plan);
int ast_fft_exec(ast_fft_plan* plan, void* in_d, void* out_d, void* stream);
int ast_fft_plan_destroy(ast_fft_plan* plan);

/* Hand-written LDS-tiled FFT passes (fp32, n in {256, 512, 1024}): the 3D R2C as
 * exactly three passes over the array (rocFFT's plan at 1024^3 runs six kernels
 * and moves ~2x the bytes).  Same results as the rocFFT plans above to fp32
 * round-off; callers fall back to rocFFT when ast_fft_tile_supported() is 0.
 *   ast_fft_tile_c2c: in-place forward transforms of length n over
 *       data[b*batch_stride + k*elem_stride + c], c < ncols contiguous columns.
 *   ast_fft_tile_rows_r2c: nrows contiguous real rows of n -> n/2+1 complex.
 *   ast_fft_tile_r2c_3d: (n,n,n) real -> (n,n,n/2+1), out = scale * sum f e^{-ikx}. */
int ast_fft_tile_supported(int dtype, size_t n);
int ast_fft_tile_c2c(void* data_d, int dtype, size_t n, size_t elem_stride, size_t ncols, size_t batch,
                     size_t batch_stride, double scale, void* stream);

/* ast_fft_tile_c2c over the k_y axis of a rank's local planes (planes_d: (nplanes, n, pitch) complex of which the first
 * ncols columns of every row are data; left intact) with ast_slab_pack fused into the stores: packed_d receives `parts`
 * blocks of (nplanes, n / parts, pitch), the send buffer of the slab transpose (pmesh's r2c transposes inside the
 * reference's FFTPower call, power_spectrum_3d.py:203-208).  parts: a power of two dividing n.  pitch >= ncols: a pitch
 * that is a multiple of 16 keeps every row piece on whole 128-byte lines (the columns ncols .. pitch - 1 are neither
 * read nor written).  With self_out_d, part self_part (the rank's own piece, which never travels) is written there
 * instead, as (nplanes, n / parts, pitch); packed_d may then be NULL when parts == 1. */
int ast_fft_tile_c2c_packed(const void* planes_d, void* packed_d, int dtype, size_t n, size_t ncols, size_t pitch,
                            size_t nplanes, int parts, int self_part, void* self_out_d, double scale, void* stream);
int ast_fft_tile_rows_r2c(const void* in_d, void* out_d, int dtype, size_t n, size_t nrows, size_t in_pitch,
                          size_t out_pitch, double scale, void* stream);
/* ast_fft_tile_rows_r2c that also leaves the low-k channel's z sums of its rows (sum_z f e^{-2 pi i kz z / n}, kz <= 6,
 * in double, from the samples the pass holds in registers anyway) at lowz_d: nrows x 7 complex128, [row][kz] - the
 * first stage of ast_lowk_modes without its second read of the planes.  fp32, n in {256, 512, 1024}. */
int ast_fft_tile_rows_r2c_lowz(const void* in_d, void* out_d, int dtype, size_t n, size_t nrows, size_t in_pitch,
                               size_t out_pitch, double scale, void* lowz_d, void* stream);
/* The z pass of a plane range of a SLAB buffer painted in stages (ast_paint_tiled_stage), folding the paint's halo records as
 * the rows are loaded - what AST_PAINT_STAGE_FOLD adds, in the same order (bit-identical): in_d = buffer plane xb0, nrows =
 * planes * n rows; halo_rec_d from ast_paint_tiled_halo for the same buffer (nx_alloc planes, not periodic in x); planes whose
 * tile row lies in [fold_row_lo, fold_row_hi) are folded here, the others must have had their FOLD stage (the rows that hold
 * ghost planes).  lowz_d: NULL, or the low-k z sums of these rows as ast_fft_tile_rows_r2c_lowz leaves them. */
int ast_fft_tile_rows_r2c_slab_halo(const void* in_d, void* out_d, int dtype, size_t n, size_t nrows, size_t out_pitch,
                                    double scale, const void* halo_rec_d, int window, int xb0, int nx_alloc, int fold_row_lo,
                                    int fold_row_hi, void* lowz_d, void* stream);
int ast_fft_tile_r2c_3d(const void* in_d, void* out_d, int dtype, size_t n, double scale, void* stream);
/* The unnormalised inverse in three tile passes (x, y, z): out_d[x] = scale * sum_k spec_k e^{+ikx} for an (n, n, n/2+1)
 * half spectrum (fp32, n in {256, 512, 1024}).  spec_d is not modified; work_d (as large as spec_d) is scratch.
 * m_hi > m_lo >= 0: only the modes with m_lo <= |m| < m_hi enter - the shell filter of the bispectrum estimator
 * (ast_shell_filter) fused into the first pass's loads; m_hi = 0: all modes. */
int ast_fft_tile_c2r_3d(const void* spec_d, void* work_d, void* out_d, int dtype, size_t n, int m_lo, int m_hi,
                        double scale, void* stream);
/* The same for up to 8 shells of one spectrum in three launches instead of 3 per shell (the x, y and z passes take the
 * shell from the launch's second grid dimension): works[i] / outs[i] - HOST arrays of device pointers, all distinct - are
 * shell i's scratch spectrum and real output, m_lo[i] < m_hi[i] (host ints) its radii.  Outputs bit-identical to the
 * single call's.  passes: 1 = the x and y passes only (works[i] then hold what the z pass reads), 2 = the z pass only, 3 = all -
 * so that a caller can run x / y shell by shell (the y pass finds the x pass's output in the Infinity Cache) and the z
 * passes of several shells in one launch (small shells fill the tails of large ones).
 * work_pitch: row pitch of the scratch spectra in complex elements, 0 or >= n/2+1 (0 = n/2+1; works[i] then hold
 * n * n * work_pitch elements).  A multiple of 16 keeps the 128-byte row pieces of the x / y passes on whole lines. */
int ast_fft_tile_c2r_3d_batch(const void* spec_d, void* const* works, void* const* outs, int dtype, size_t n,
                              const int* m_lo, const int* m_hi, int count, double scale, int passes, size_t work_pitch,
                              void* stream);
/* a-4 + a-5 fused: FFTPower's shell sums of an (n, n, n) real grid.  z and y passes
 * go through scratch_d (line-aligned row pitch); the x pass adds w |delta_k|^2 of its
 * modes to per-workgroup shell tables instead of storing delta_k, and a fixed-order
 * reduction adds them into psum_d (same meaning as in ast_power_bin_1d; auto power). */
/* The bispectrum estimator's last two steps in ONE kernel: the z passes (C2R) of all shells - works[s] = shell s's scratch
 * spectrum after the masked x and y passes (ast_fft_tile_c2r_3d_batch, passes = 1), (n, n, work_pitch) complex64, m_hi[s] its
 * outer radius - and the triangle sums out_d[t] = sum_x f_a f_b f_c, f_s = scale * C2R_z(works[s]), for the ntri triangles
 * tri_d[(ntri, 3)] of shell slots.  One workgroup transforms one (x, y) row of EVERY shell and forms the sums from LDS: the
 * real cubes (31 x 0.5 GB written and read back at 512^3) never reach HBM.  bispectrum_3d.py:165-215 holds no such
 * arithmetic; this is the estimator its docstring cites.  nshells <= 32; ntri <= 512 (1024 at n = 1024). */
size_t ast_fft_tile_c2r_triangles_scratch_bytes(void);
int ast_fft_tile_c2r_triangles(void* const* works, const int* m_hi, int nshells, int dtype, size_t n, size_t work_pitch,
                               double scale, const int* tri_d, int ntri, void* scratch_d, double* out_d, void* stream);
size_t ast_fft_tile_power_scratch_bytes(size_t n);
/* test hook: out_d[v] = the fused binning's floor(sqrt(v)) (one hardware sqrt, no repair), v < count */
int ast_fft_tile_isqrt_table(int* out_d, int count, void* stream);
/* `binning`: AST_BIN_INTEGER / AST_BIN_FLOAT64, see ast_power_bin_1d. */
#define AST_BIN_INTEGER 0
#define AST_BIN_FLOAT64 1
/* `mean`: a constant subtracted from every cell as it is loaded (0 = none).  It only
 * changes the DC mode, which FFTPower discards, but with it the fp32 round-off of all
 * other modes no longer scales with the O(1) mean density (cold low-k shells gain). */
/* `lowk` = 1: the modes with |m_i| <= 5 are ALSO evaluated as DFT sums in double (one more read of the grid) and the
 * five lowest shells, |m| in [1, 6), take their sums from there: the fp32 transform's white round-off floor
 * (~1e-7 of the rms amplitude per mode) otherwise limits shells that hold 1e-5 of the peak power to ~2e-6 / |m|^2. */
int ast_fft_tile_power_3d(const void* grid_d, void* scratch_d, size_t scratch_bytes, int dtype, size_t n,
                          double boxsize, double mean, int lowk, int binning, double* psum_d, void* stream);
/* The same for a grid painted with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD (fp32, whole periodic grid):
 * halo_rec_d comes from ast_paint_tiled_halo on the paint's workspace; the records are added to the border
 * rows as the z pass loads them, in the order the paint's own fold kernel uses (bit-identical result). */
int ast_fft_tile_power_3d_halo(const void* grid_d, const void* halo_rec_d, int window, void* scratch_d,
                               size_t scratch_bytes, int dtype, size_t n, double boxsize, double mean,
                               int lowk, int binning, double* psum_d, void* stream);

/* The axis-0 pass of a slab-decomposed transform fused with the shell binning (the multi-GPU counterpart of the fused x
 * pass): block_d is the rank's (n, nloc, pitch) block of the spectrum after the all-to-all - all k_x, k_y = ky0 ..
 * ky0 + nloc - 1, half k_z, `pitch` >= n/2+1 complex per row.  psum_d[shell] += L^3 sum w |scale * X|^2 over the block's
 * modes; shells below first_bin are skipped (low-k channel); the block's contents afterwards are undefined.
 * scratch_d: ast_fft_tile_block_power_scratch_bytes(n, nloc). */
size_t ast_fft_tile_block_power_scratch_bytes(size_t n, size_t nloc);
int ast_fft_tile_block_power(void* block_d, void* scratch_d, size_t scratch_bytes, int dtype, size_t n, size_t nloc,
                             size_t ky0, size_t pitch, double scale, double boxsize, int first_bin, int binning,
                             double* psum_d, void* stream);

/* DISC LAYOUT of the half spectrum between the k_y pass and the last pass: only what FFTPower keeps.
 * FFTPower(mode="1d", kmin=k_F) drops every mode with |m| >= N/2 (power_spectrum_3d.py:189-195), so after the k_y pass a row
 * (k_y, 16-column k_z tile starting at k_z0) with k_y^2 + k_z0^2 > (N/2)^2 holds nothing any shell takes (the edge itself
 * stays: under AST_BIN_FLOAT64 a vector of norm exactly N/2 may fall into the last shell) - 21.5 % of the half plane.
 * The k_y rows are dealt to `parts` owners in blocks of r1 rows (32 at n = 1024, else 16), balanced by the area of the disc
 * they cover (contiguous ranges would leave the owners of the rows around k_y = 0 with full planes); a part's plane holds,
 * tile by tile, its rows that reach into the disc, 16 complex each.  The slab transpose sends these planes (pmesh transposes
 * inside the reference's one FFTPower call; nothing in the reference corresponds to the layout).  fp32, n in {256, 512,
 * 1024}, parts dividing n / r1.
 *   ast_fft_tile_disc_layout: plane_elems[parts] = complex elements per plane of each part, block_part[n / r1] = owner of
 *     every row block (or NULL), *r1_out = rows per block.  Host arithmetic only.
 *   ast_fft_tile_disc_table: the table [tile][row block] x (offb, S, cumS, lo | hi << 8 | part << 16) as int32 (host).
 *   ast_fft_tile_c2c_disc: the k_y pass of planes_d ((nplanes, n, pitch) complex, left intact) storing part q's rows of
 *     plane b at packed_d + nplanes * cumS[q] + b * S[q], the part self_part at self_out_d + b * S[self_part] (its place in
 *     the rank's own block; self_part < 0 and self_out_d NULL: none; packed_d may be NULL when parts == 1).
 *   ast_fft_tile_disc_block_power: the last pass over part `part`'s block ((n, S[part]) complex) fused with the shell
 *     binning, psum_d[shell] += L^3 sum w |scale * X|^2; shells below first_bin are skipped. */
int ast_fft_tile_disc_layout(size_t n, int parts, unsigned* plane_elems, unsigned char* block_part, int* r1_out);
int ast_fft_tile_disc_table(size_t n, int parts, int* out, size_t out_ints);
int ast_fft_tile_c2c_disc(const void* planes_d, void* packed_d, int dtype, size_t n, size_t pitch, size_t nplanes, int parts,
                          int self_part, void* self_out_d, double scale, void* stream);
size_t ast_fft_tile_disc_power_scratch_bytes(size_t n, int parts);
int ast_fft_tile_disc_block_power(void* block_d, void* scratch_d, size_t scratch_bytes, int dtype, size_t n, int parts, int part,
                                  double scale, double boxsize, int first_bin, int binning, double* psum_d, void* stream);

/* ------------------------------------------------------------- communication (ast_comm_init, ast_slab_transpose)
 * The multi-GPU exchange steps for a caller that binds this library alone: RCCL over xGMI, one process per GPU.  (The
 * Python side of this repository reaches RCCL through torch.distributed and does not call these.)  RCCL is loaded lazily
 * at the first call; the library has no link-time dependency on it.  The reference has no counterpart: pmesh / pfft
 * transpose over MPI inside the one FFTPower call (power_spectrum_3d.py:189-195).
 *   ast_comm_unique_id: 128 bytes from ncclGetUniqueId on ONE rank, to be handed to the others by the caller's own means.
 *   ast_comm_init / ast_comm_destroy: a communicator of `nranks` ranks for the current device.
 *   ast_slab_transpose: one group of point-to-point operations - to every peer q send_count[q] REAL elements of `dtype`
 *     (a complex value is two) from send_d + send_offset[q], from every peer recv_count[q] into recv_d + recv_offset[q];
 *     a piece addressed to the rank itself is copied on the stream.  With the disc layout of ast_fft_tile_c2c_disc for
 *     `nplanes` local planes: send_offset[q] = 2 nplanes cumS[q], send_count[q] = 2 nplanes S[q]; recv_offset[q] =
 *     2 (q nloc + p0) S[rank], recv_count[q] = 2 nplanes S[rank].  Asynchronous on `stream`.
 *   ast_comm_allreduce_sum: in-place sum of float64 values over the ranks (shell sums, low-k modes). */
typedef struct ast_comm ast_comm;
int ast_comm_unique_id(void* id_out, size_t id_bytes);
int ast_comm_init(ast_comm** out, int nranks, int rank, const void* id, size_t id_bytes);
int ast_comm_destroy(ast_comm* comm);
int ast_slab_transpose(ast_comm* comm, const void* send_d, const size_t* send_offset, const size_t* send_count, void* recv_d,
                       const size_t* recv_offset, const size_t* recv_count, int dtype, void* stream);
int ast_comm_allreduce_sum(ast_comm* comm, double* buf_d, size_t count, void* stream);

/* The low-k channel as separate calls, for slab-decomposed grids: every rank adds the contribution of its own
 * planes to the (2*6+1)^2 * 7 modes |m_i| <= 6, m_z >= 0 (complex128, [kx + 6][ky + 6][kz]); the modes are summed over
 * ranks; the sums of the ast_lowk_shell_count() lowest shells are then taken from them.
 *   planes_d: nx complete planes (x0 .. x0 + nx - 1 of the global axis 0) of n x n fp32 cells; n in {256, 512, 1024}.
 *   work_d: ast_lowk_work_bytes(n, nx) bytes.  accumulate: add into modes_d instead of overwriting it. */
size_t ast_lowk_work_bytes(size_t n, size_t nx);
int ast_lowk_mode_count(void);
int ast_lowk_shell_count(void);
int ast_lowk_modes(const void* planes_d, int dtype, size_t n, size_t x0, size_t nx, int accumulate, void* modes_d,
                   void* work_d, size_t work_bytes, void* stream);
/* The same from z sums that ast_fft_tile_rows_r2c_lowz has already left at the START of work_d ([plane][y][kz] for the
 * nx planes x0 .. x0 + nx - 1, any plane ranges in any order): the y and x sums only, no read of the planes. */
int ast_lowk_modes_from_z(size_t n, size_t x0, size_t nx, int accumulate, void* modes_d, void* work_d, size_t work_bytes,
                          void* stream);
/* sums_d[s] = L^3 sum w |modes / n^3|^2 over the modes of shell s (membership by `binning`), s < ast_lowk_shell_count(). */
int ast_lowk_shell_sums(const void* modes_d, size_t n, double boxsize, int binning, double* sums_d, void* stream);

/* ---------------------------------------------- a-5: k-shell power binning */

/* FFTPower(mode="1d", dk = kmin = 2 pi / L) shell sums over a block of the
 * half spectrum.  Replaces nbodykit project_to_basis as reached from
 * power_spectrum_3d.py:189-224 and stats_subfind.py:142-150.
 *   spec1_d, spec2_d: interleaved complex, dims (i0_count, i1_count,
 *       nmesh/2+1), C order, holding pmesh-normalised delta_k (spec2_d NULL =
 *       auto spectrum).  Global integer frequencies of the first two axes
 *       start at i0_start / i1_start (single GPU: 0, nmesh, 0, nmesh).
 *   ksum_d, psum_d (double) and nmodes_d (int64), nmesh/2-1 bins each, are
 *       ACCUMULATED into (caller zero-fills):  sum w |k|,
 *       sum w Re(d1 conj(d2)) L^3,  sum w,  with Hermitian weight w = 2 for
 *       0 < i2 < nmesh/2 else 1.  Shell of a mode: floor(|m|) - 1; the DC mode and |m| >= nmesh/2
 *       are dropped.
 *   binning: how lattice vectors of EXACTLY integer norm (perfect-square |m|^2, which sit on a shell
 *       edge) are assigned.  AST_BIN_INTEGER: to the shell they open, floor(|m|) - 1 (exact integer
 *       arithmetic, independent of L).  AST_BIN_FLOAT64: as nbodykit's float64 expressions round -
 *       digitize(kx^2 + ky^2 + kz^2, kedges^2) with k_i = (2 pi / L) m_i, kedges = arange(k_F, ..., k_F) -
 *       which puts some of them one shell lower (and some |m| = nmesh/2 vectors into the last shell),
 *       depending on L; this is the reference's behaviour as far as nbodykit's published source fixes it
 *       (nbodykit is un-vendored: parity unpinned).  All other vectors are unaffected.
 *   k = ksum/nmodes, P = psum/nmodes is left to the caller so that slab
 *       partials can be summed first.
 *   ksum_d / nmodes_d depend only on (nmesh, L, block), not on the data: pass
 *       both NULL to skip them (callers cache them), or spec1_d = psum_d = NULL
 *       to compute only them. */
int ast_power_bin_1d(const void* spec1_d, const void* spec2_d, int dtype, int nmesh,
                     double boxsize, int i0_start, int i0_count, int i1_start, int i1_count,
                     double* ksum_d, double* psum_d, long long* nmodes_d, int binning, void* stream);

/* --------------------------------------------------- a-10: bispectrum */

/* The reference's Bispectrum3D computes P(k) (bispectra/bispectrum_3d.py:165-215);
 * these two kernels carry the FFT (Scoccimarro) estimator its docstring cites
 * (:42-44): delta_i(x) = IFFT[delta_k 1(k in shell i)], I_i(x) = IFFT[1(k in i)],
 * B(i,j,l) = L^6 sum_x d_i d_j d_l / sum_x I_i I_j I_l.
 *
 * out = in * 1[m_lo <= |m| < m_hi] (or just the indicator when in_d is NULL),
 * exact integer comparison of |m|^2, over the same spectrum block layout as
 * ast_power_bin_1d. */
int ast_shell_filter(const void* in_d, void* out_d, int dtype, int nmesh, int m_lo, int m_hi,
                     int i0_start, int i0_count, int i1_start, int i1_count, void* stream);
/* I_s(x) = sum_{k in shell} e^{ikx} WITHOUT an inverse transform (bispectra/bispectrum_3d.py:42-44, the estimator's
 * triangle counts): the shell indicator m_lo <= |m| < m_hi on the full lattice as an (n, n, n) float64 array
 * (ast_shell_mask_real) is real and even, so its forward transform (ast_fft64_r2c_3d, scale 1) IS I_s on the half
 * lattice, real up to round-off; ast_half_real_to_full unfolds the real part onto the full (n, n, n) lattice. */
int ast_shell_mask_real(double* out_d, int nmesh, int m_lo, int m_hi, void* stream);
int ast_half_real_to_full(const void* spec_d, double* out_d, int nmesh, void* stream);

/* *out_d += sum_i a[i] * b[i] * c[i]  (double accumulator, device). */
int ast_triple_product_sum(const void* a_d, const void* b_d, const void* c_d, int dtype,
                           size_t count, double* out_d, void* stream);

/* out_d[t] = sum_i f[tri[3t]][i] * f[tri[3t+1]][i] * f[tri[3t+2]][i] for ntri <= 256 triangles over nfields device
 * arrays of `count` reals (fields_d: device array of nfields device pointers; tri_d: device int32 [ntri][3] of field
 * indices): every field is read ONCE (LDS-staged chunks), not once per triangle - the 75 cube sums of a 512^3
 * bispectrum (bispectrum_3d.py:42-44's estimator) in one pass.  Double products and sums in a fixed order
 * (deterministic).  scratch_d: ast_triple_product_sums_scratch_bytes() bytes.  nfields * 257 * sizeof(real) <= 160 KB. */
size_t ast_triple_product_sums_scratch_bytes(void);
int ast_triple_product_sums(const void* const* fields_d, int nfields, int dtype, size_t count, const int* tri_d,
                            int ntri, void* scratch_d, double* out_d, void* stream);

/* ------------------------------------------------- slab transpose helpers */

/* Pack the (n0, n1, n2) complex block so that the n1 axis is split into
 * `parts` equal chunks, each stored contiguously as (n0, n1/parts, n2):
 * the send layout of the slab all-to-all.  unpack is the inverse for the
 * receive side: `parts` blocks of (n0, n1, n2) -> one (parts*n0, n1, n2)
 * array is already contiguous, so only pack is needed on the way out and
 * this inverse on the way back (c2r). */
int ast_slab_pack(const void* in_d, void* out_d, int dtype, size_t n0, size_t n1, size_t n2,
                  int parts, void* stream);
int ast_slab_unpack(const void* in_d, void* out_d, int dtype, size_t n0, size_t n1, size_t n2,
                    int parts, void* stream);

/* ------------------------------------------------------ a-7: kappa stack */

/* out[i] = sum_p planes[p][i] * wnum[p] / wden[p], added in plane order
 * p = 0..nplanes-1 exactly like the reference's running `+=`
 * (rays/rayramses.py:224-232, simcoll.py:322-336) with the lensing-kernel
 * re-weighting quantity * g(x_mid, x_s') / g(x_mid, x_s) of
 * rayramses.py:306-312 / simcoll.py:424-430.  planes_d: device array of
 * nplanes device pointers; wnum_d / wden_d: device doubles or NULL (no
 * re-weighting).  fp64 results are bit-identical to numpy.  aligned16 != 0: the caller
 * promises that every plane pointer is 16-byte aligned (the pointers live on the device and are
 * not inspected); the kernel then moves 16 bytes per lane when `count` divides evenly. */
int ast_kappa_stack(const void* const* planes_d, const double* wnum_d, const double* wden_d,
                    int nplanes, size_t count, int dtype, void* out_d, int aligned16, void* stream);

/* --------------------------------------- a-8 / a-9: per-map kappa pipeline */

typedef struct ast_lens_plan ast_lens_plan;

/* Plan for kappa -> (alpha1, alpha2) / phi on an nc x nc map of side bsz
 * [rad]: owns the (2nc)^2 rocFFT plans, the padded work arrays and the cached
 * spectra of the isotropic kernels of lensing_funcs.c:45-83,117-148. */
int ast_lens_plan_create(ast_lens_plan** plan, int nc, double bsz);
int ast_lens_plan_destroy(ast_lens_plan* plan);
/* FFTPower of an in-memory float64 grid (the reference's dtype: power_spectrum_3d.py:183-224 on float64 arrays) for
 * n = 128 ... 2048 (powers of two) through hand-written double-precision passes (z rows, then ONE strided pass per axis, the last
 * one fused with the shell binning): psum_d[shell] += L^3 sum_modes w |delta_k|^2, delta_k = rfftn(grid) / n^3.
 * grid_d is not modified; scratch_d: ast_fft64_power_scratch_bytes(n) bytes. */
int ast_fft64_supported(size_t n);
size_t ast_fft64_power_scratch_bytes(size_t n);
int ast_fft64_power_3d(const double* grid_d, void* scratch_d, size_t scratch_bytes, size_t n, double boxsize, int binning,
                       double* psum_d, void* stream);
/* ast_fft64_power_3d of a SINGLE-precision grid, transformed in double (the z pass widens the rows as it loads them): fp32
 * cubes the fp32 tile passes do not cover (n = 128, 2048) without a float64 copy of the grid.  Same scratch. */
int ast_fft64_power_3d_f32(const float* grid_d, void* scratch_d, size_t scratch_bytes, size_t n, double boxsize, int binning,
                           double* psum_d, void* stream);
/* The same shell sums for a SINGLE-precision cube of side 2048 with every pass in single precision (three register FFTs
 * per axis; the fp32 tile passes end at 1024): psum_d[shell] += L^3 sum w |rfftn(grid - mean) / n^3|^2, FFTPower's sums
 * (power_spectrum_3d.py:189-224).  `mean` is subtracted as the rows are loaded (only the discarded DC mode sees it).  The
 * lowest shells carry fp32 round-off; callers patch them from ast_lowk_modes / ast_lowk_shell_sums (device.py does).
 * Half the bytes of ast_fft64_power_3d_f32 on every pass. */
int ast_fft32_big_supported(size_t n);
size_t ast_fft32_big_power_scratch_bytes(size_t n);
int ast_fft32_big_power_3d(const float* grid_d, void* scratch_d, size_t scratch_bytes, size_t n, double boxsize, int binning,
                           double mean, double* psum_d, void* stream);
/* ... of a grid painted with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD: halo_rec_d (ast_paint_tiled_halo) is folded into the
 * border rows as the z rows and the low-k sums load them (the paint's fold kernel - 5 ms at side 2048 - is not run) */
int ast_fft32_big_power_3d_halo(const float* grid_d, const float* halo_rec_d, int window, void* scratch_d, size_t scratch_bytes,
                                size_t n, double boxsize, int binning, double mean, double* psum_d, void* stream);
/* spec_d (n, n, n / 2 + 1) complex double, contiguous = rfftn(grid_d) * scale (pmesh's r2c with scale = 1 / n^3) through
 * the same passes; grid_d is not modified */
int ast_fft64_r2c_3d(const double* grid_d, void* spec_d, size_t n, double scale, void* stream);
/* the same for a grid left by ast_paint_tiled(AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD): halo_rec_d
 * (ast_paint_tiled_halo) is folded into the border rows as the z pass loads them */
int ast_fft64_power_3d_halo(const double* grid_d, const double* halo_rec_d, int window, void* scratch_d, size_t scratch_bytes,
                            size_t n, double boxsize, int binning, double* psum_d, void* stream);

/* Column transforms of the zero-padded lens convolution (lensing_funcs.c:85-115 + fft_convolve.c:60-90), double,
 * hand-written two-pass (four-step) passes over data_d[len][pitch] complex that skip the half known to be zero.
 * len in {256 .. 8192}, powers of two (ast_lens_cols_supported).  The forward transform (in place) reads rows
 * < nonzero_rows only (len or len / 2) and leaves frequency k = k1 + N1 k2 at row N2 k1 + k2 (a fixed permutation;
 * spectra of one length are only ever multiplied with each other and transformed back).  The inverse
 * (unnormalised) transforms spec_d * mul_d (mul_d NULL: spec_d alone) into out_d and writes the first keep_rows rows
 * (len or len / 2), in natural order. */
int ast_lens_cols_supported(size_t len);
/* The row transforms of the same convolution for nc = 128 .. 4096, powers of two (ast_lens_rows_supported): forward - row r < nc of
 * spec_d (pitch complex per row) = the length-2nc R2C of (kappa_d[r][0 .. nc), nc zeros), kappa read unpadded
 * (zero_padding, lensing_funcs.c:8-19, never materialised); inverse - out_d[r][0 .. nc) = scale * the first nc reals of
 * the unnormalised length-2nc C2R of spec_d[r][0 .. nc] (corner_matrix, lensing_funcs.c:33-43, and the
 * 1 / (nx ny) dx dy of fft_convolve.c:88 in the store). */
int ast_lens_rows_supported(size_t nc);
int ast_lens_rows_forward(const double* kappa_d, size_t nc, void* spec_d, size_t pitch, void* stream);
/* The same for FULL rows of 2 nc reals (no zero padding), nrows of them with pitch 2 nc: the rows of the convolution
 * kernels (lensing_funcs.c:45-83), so that a lens plan of a supported size runs without rocFFT. */
int ast_lens_rows_forward_full(const double* in_d, size_t nc, size_t nrows, void* spec_d, size_t pitch, void* stream);
int ast_lens_rows_inverse(const void* spec_d, size_t pitch, size_t nc, double scale, double* out_d, void* stream);
int ast_lens_cols_forward(void* data_d, size_t len, size_t pitch, size_t ncols, size_t nonzero_rows, void* stream);
int ast_lens_cols_inverse(const void* spec_d, const void* mul_d, void* out_d, size_t len, size_t pitch, size_t ncols,
                          size_t keep_rows, void* stream);
/* The whole column part of a convolution with nmul (1 or 2) kernel spectra mul_d[m] (permuted order, as
 * ast_lens_cols_forward leaves them): out_d[m] = the first keep_rows rows of IFFT_cols(FFT_cols(data_d) * mul_d[m]),
 * unnormalised, natural order.  data_d is overwritten (forward pass A in place); the second forward pass, the products
 * and the first inverse pass are one kernel, so the spectrum of data_d never goes to memory (3 array passes fewer for
 * kappa -> alpha1, alpha2 than forward + 2 x inverse).  muls / outs are host arrays of device pointers. */
int ast_lens_cols_convolve(void* data_d, size_t len, size_t pitch, size_t ncols, size_t nonzero_rows, const void* const* muls,
                           void* const* outs, int nmul, size_t keep_rows, void* stream);

/* Device-pointer variants (fp64, C-contiguous nc*nc). */
int ast_kappa_to_alphas(ast_lens_plan* plan, const double* kappa_d, double* alpha1_d,
                        double* alpha2_d, void* stream);
int ast_kappa_to_phi(ast_lens_plan* plan, const double* kappa_d, double* phi_d, void* stream);

/* libglsg.so-compatible entry points (rays/skys/lib_so_cgls/lensing_funcs.h:5,7;
 * bound by ctypes at rays/skys/sky_utils.py:402-435): HOST pointers,
 * caller-allocated outputs, synchronous, void return like the original. */
void kappa0_to_alphas(double* kappa0, int Nc, double bsz, double* alpha1, double* alpha2);
void kappa0_to_phi(double* kappa0, int Nc, double bsz, double* phi);

typedef struct ast_smooth_plan ast_smooth_plan;

/* Gaussian smoothing of an npix x npix fp64 map (Filters.gaussian,
 * rays/utils/filters.py:181-225 -> lenstools ConvergenceMap.smooth):
 *   mode 0 "gaussianFFT": irfft2(exp(-0.5 l^2 (2 pi sigma_px)^2) rfft2(img)),
 *          l from (r)fftfreq(npix) — periodic.
 *   mode 1 "gaussian": separable real-space kernel, reflect boundary,
 *          truncate 4 sigma (scipy.ndimage.gaussian_filter).
 *   mode 2: as mode 1 with scipy's "mirror" boundary (d c b | a b c d | c b a, the edge
 *          pixel not repeated) - what skimage.transform.resize(mode="reflect", numpy's
 *          naming) hands to ndimage for its anti-aliasing prefilter (SkyArray.resize,
 *          rays/skys/sky_array.py:475-496).
 * In place on img_d. */
int ast_smooth_plan_create(ast_smooth_plan** plan, int npix);
int ast_smooth_plan_destroy(ast_smooth_plan* plan);
int ast_gaussian_smooth(ast_smooth_plan* plan, double* img_d, double sigma_px, int mode,
                        void* stream);

/* Flat-sky angular power spectrum binning (power_spectra/angular_power_spectrum.py:38-53 -> lenstools
 * ConvergenceMap.powerSpectrum, restated): ft1_d / ft2_d (or NULL = auto) are rfft2 half planes (npix x (npix/2+1),
 * complex128, unnormalised); pixel (i, j) has |l| = 2 pi / angle * sqrt(min(i, npix-i)^2 + j^2) and falls into bin k
 * when edges[k] < |l| <= edges[k+1].  psum_d[k] += sum Re(ft1 conj ft2), hits_d[k] += pixels (device uint64);
 * P(k) = psum / hits * (angle / npix^2)^2 on the host.  edges_d: nbins + 1 doubles on the device, ascending. */
int ast_flat_power_bin(const void* ft1_d, const void* ft2_d, int npix, double angle_rad, const double* edges_d,
                       int nbins, double* psum_d, unsigned long long* hits_d, void* stream);
/* out = in * 1[l_lo < |l| <= l_hi] on the same half plane (in_d = NULL: the indicator itself) - the ring fields of the
 * FFT estimator of the equilateral flat-sky bispectrum (bispectra/bispectrum_2d.py:33-50). */
int ast_ring_filter_2d(const void* in_d, void* out_d, int npix, double angle_rad, double l_lo, double l_hi, void* stream);

/* Local maxima of an npix x npix map (SkyArray.wl_peak_counts, rays/skys/sky_array.py:435-472 ->
 * lenstools ConvergenceMap.locatePeaks): interior pixels strictly larger than their 8 neighbours with a
 * value in [lo, hi).  Heights go to values_d (dtype), flat row-major pixel indices to index_d, at most
 * `cap` of each; *count_d (device uint64) receives the number FOUND (may exceed cap).  The order in the
 * output arrays is not defined (sort by index for lenstools' scan order). */
int ast_peak_find(const void* img_d, int dtype, int npix, double lo, double hi, size_t cap, void* values_d,
                  long long* index_d, unsigned long long* count_d, void* stream);

/* Exact order statistics: out_host[j] = the ks_host[j]-th smallest element (0-based) of buf_d - what
 * np.percentile interpolates between (sky_array.py:452-457).  Radix select on order-preserving keys, six
 * passes over the buffer per k; SYNCHRONOUS (host arrays in and out).  scratch_d: 2048 uint64 on the device. */
int ast_order_statistics(const void* buf_d, int dtype, size_t count, const size_t* ks_host, int nk,
                         double* out_host, unsigned long long* scratch_d, void* stream);

/* Sum of a buffer in double, fixed summation order (bit-reproducible): out_d[0]; out_d must hold
 * 1 + AST_SUM_PARTS doubles (the rest is scratch). */
#define AST_SUM_PARTS 1024
int ast_sum(const void* buf_d, int dtype, size_t count, double* out_d, void* stream);

/* min and max of a buffer -> out_d[0], out_d[1] (double, device). */
int ast_minmax(const void* buf_d, int dtype, size_t count, double* out_d, void* stream);

/* np.histogram(data, bins=nbins, range=(lo, hi)) counts: uniform bins,
 * right-most bin closed (SkyArray.pdf, rays/skys/sky_array.py:428-433).
 * counts_d (int64, nbins) is ACCUMULATED into. */
int ast_histogram(const void* buf_d, int dtype, size_t count, double lo, double hi, int nbins,
                  long long* counts_d, void* stream);

/* np.histogram(buf, bins=nbins) with range=None (sky_array.py's PDF calls) without a host round trip between the
 * min/max pass and the counting pass: range_d[0..1] receives {min, max} (a degenerate range counts against
 * [min - 0.5, max + 0.5] like numpy), counts_d (zeroed by the caller) the bin counts. */
int ast_histogram_auto(const void* buf_d, int dtype, size_t count, int nbins, long long* counts_d, double* range_d,
                       void* stream);

/* out[i] = a[i] + b[i]  (add_galaxy_shape_noise, sky_array.py:693-706). */
int ast_add(const void* a_d, const void* b_d, void* out_d, int dtype, size_t count, void* stream);

/* ------------------------------------- f-2: analytic NFW halo signals (next row) */

/* Paint the deflection-angle (signal 0) or moving-lens temperature (signal 1) stamp
 * of every halo onto an npix x npix fp64 map, clipped at the map edge.  Replaces
 * SkyUtils.analytic_Halo_signal_to_SkyArray and the NFW_*_map / add_patch_to_map
 * helpers it loops over (rays/skys/sky_utils.py:79-282).  Per-halo device arrays:
 * r200 [deg], m200 [M_sun], concentration, angular diameter distance [Mpc] (the
 * reference passes Dc * 0.6774), transverse velocities [km/s] (signal 1 only), stamp
 * edge length int(2 * r200_pix * extent) + 1, stamp centre (theta1_pix = column,
 * theta2_pix = row).  dir_mask: bit 0 = direction 0 (x), bit 1 = direction 1 (y).
 * map_d is ACCUMULATED into. */
int ast_nfw_paint(const double* r200_deg_d, const double* m200_d, const double* c_nfw_d, const double* dist_d,
                  const double* vel_x_d, const double* vel_y_d, const int* stamp_npix_d, const int* cen_x_d,
                  const int* cen_y_d, size_t nhalo, double extent, int dir_mask, int suppress,
                  double suppression_r, int signal, double* map_d, int npix, void* stream);

/* limg[y, x] += simg[i, j] for the in-bounds part of a stamp centred at (cen_x, cen_y)
 * (SkyUtils.add_patch_to_map, sky_utils.py:140-173). */
int ast_add_patch(double* limg_d, int nl, const double* simg_d, int ns, int cen_x, int cen_y, void* stream);

/* --------------------------- f-3: flat-sky window filters (next row, first part) */

/* Filters.gaussian_third_derivative (order 3, "DGD3") and gaussian_first_derivative (order 1)
 * of rays/utils/filters.py:305-400: out = img * d^order/d(axis)^order W, W a sum of Gaussians of
 * the pixel distance to the map centre (sigma_pix = ceil(npix * theta_i / theta)), derivatives by
 * repeated np.gradient(.., h, edge_order=2) with h = theta_fov / npix.  work_d: 2 npix^2 doubles. */
int ast_dgd_filter(const double* img_d, double* out_d, double* work_d, int npix, double sigma_pix, double h,
                   int axis, int order, void* stream);

/* Filters.apodization (filters.py:150-178): img * outer(hann(npix), hann(npix)). */
int ast_hann_apodize(const double* img_d, double* out_d, int npix, void* stream);

/* SkyArray.resize (sky_array.py:475-496: skimage.transform.resize(img, (nout, nout), anti_aliasing=True), "lower the
 * nr. of pixels"), resampling half: scipy.ndimage.zoom(img, nout / nin, order=1, grid_mode=True) - bilinear samples at
 * (o + 1/2) nin / nout - 1/2 - for nout <= nin, fp64, out of place.  The anti-aliasing half is ast_gaussian_smooth
 * (mode 1, sigma = (nin / nout - 1) / 2) on the input first.  scikit-image is not pinned by the reference's lock file:
 * this is the algorithm of scikit-image >= 0.19 (Gaussian prefilter + ndimage.zoom), restated. */
int ast_zoom_linear(const double* img_d, int nin, double* out_d, int nout, void* stream);

/* SkyUtils.convert_deflection_to_shear (rays/skys/sky_utils.py:342-362, called by SkyArray.convert_deflection_to_shear,
 * sky_array.py:820-849): gamma1 = 0.5 ((1 - d0 alpha1) - (1 - d1 alpha2)), gamma2 = 0.5 (-d0 alpha2 - d1 alpha1) with
 * d0 / d1 = np.gradient along axis 0 / 1 (edge_order 1) at uniform spacing h.  The reference's body is unfinished (its
 * `coord` is undefined and the call site passes one array): h is the pixel size, opening_angle / npix, in the unit of
 * alpha.  npix x npix fp64 maps, out of place; bit-identical to the numpy expressions. */
int ast_deflection_to_shear(const double* alpha1_d, const double* alpha2_d, int npix, double h, double* gamma1_d,
                            double* gamma2_d, void* stream);

/* scipy.ndimage.gaussian_filter(img, sigma, order=(order0, order1), mode) as called by
 * Filters.gaussian_third_derivative_convolution (filters.py:260-304): mode 0 "reflect", 1 "nearest".
 * work_d: npix^2 + 2 (2 r + 1) doubles, r = int(4 sigma + 0.5). */
int ast_gaussian_filter_order(const double* img_d, double* out_d, double* work_d, size_t work_doubles, int npix,
                              double sigma, int order0, int order1, int mode, void* stream);

/* scipy.ndimage.convolve(img, window) with its defaults (reflect, origin 0), the last line of
 * Filters.gaussian_compensated (filters.py:415-459).  img_d != out_d. */
int ast_convolve2d(const double* img_d, const double* window_d, double* out_d, int npix, int kh, int kw,
                   void* stream);

/* Filters.aperture_photometry (filters.py:40-73): out = img - mean(img[alpha_pix < d < sqrt(2) alpha_pix]).
 * work_d: 2048 doubles. */
int ast_aperture_photometry(const double* img_d, double* out_d, double* work_d, int npix, double alpha_pix,
                            void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ASTRILD_HIP_H */
