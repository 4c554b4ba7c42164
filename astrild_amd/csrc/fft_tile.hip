// a-4, hand-written: LDS-tiled power-of-two FFT passes for the 3D R2C transform
// (fp32, N in {256, 512, 1024}).
//
// rocFFT's 3D R2C at 1024^3 runs six kernels (two of them transposes) and moves
// ~2.1x the algorithmic bytes (profiles/r01_pmc_traffic.json).  Here the
// transform is exactly three passes, each reading and writing the array once:
//
//   z  rows_r2c_kernel     contiguous real rows -> half-spectrum rows (N real ->
//                          N/2 complex FFT + the even/odd split, all in LDS)
//   y  strided_c2c_kernel  for every x-plane: columns of 16 adjacent kz, N long
//   x  strided_c2c_kernel  same kernel, element stride N * (N/2+1)
//
// Each workgroup does a length-N transform of a tile of C columns as
// N = R1 * R2: R1-point FFTs in registers straight from global memory, twiddle,
// ONE round trip through LDS, R2-point FFTs in registers, store.  Loads/stores
// are C * 8 B = 128 B contiguous per row of the tile; every thread keeps R1 (R2)
// independent 8-byte loads in flight.
#include "ast_common.h"
#include "paint_tile_geom.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

// cos / sin (2 pi k / 32), k = 0..15
__device__ constexpr float kCos32[16] = {
    1.0f, 0.9807852804032304f, 0.9238795325112867f, 0.8314696123025452f, 0.7071067811865476f,
    0.5555702330196023f, 0.38268343236508984f, 0.19509032201612833f, 0.0f, -0.1950903220161282f,
    -0.3826834323650897f, -0.555570233019602f, -0.7071067811865475f, -0.8314696123025453f,
    -0.9238795325112867f, -0.9807852804032304f};
__device__ constexpr float kSin32[16] = {
    0.0f, 0.19509032201612825f, 0.3826834323650898f, 0.5555702330196022f, 0.7071067811865475f,
    0.8314696123025452f, 0.9238795325112867f, 0.9807852804032304f, 1.0f, 0.9807852804032304f,
    0.9238795325112867f, 0.8314696123025455f, 0.7071067811865476f, 0.5555702330196022f,
    0.3826834323650899f, 0.1950903220161286f};

// Complex numbers as 2-wide vectors: an add / sub is ONE v_pk_add_f32 on the number's own register
// pair and a multiply two packed instructions (v_pk_mul_f32 + v_pk_fma_f32 with lane selects).  Written
// component-wise on float2 the compiler also packs, but across DIFFERENT numbers, and pays for it with
// one v_mov_b32 per operand (439 moves next to 592 packed instructions in the 1024-point pass).
#ifndef FFT_NT
#define FFT_NT 3                    // 1 nontemporal loads, 2 nontemporal stores, 3 both
#endif
// Nontemporal accesses for arrays that are streamed through once per pass and are far larger than the caches: the three
// forward passes at N = 1024 (4.4 GB: z 2.15 -> 2.11, y 2.06 -> 2.00, x 1.28 -> 1.19 ms).  NOT for smaller cubes or the
// inverse passes of the bispectrum, which read one 0.5 GB spectrum 31 times out of the Infinity Cache (19.6 -> 22.3 ms with them).
typedef float f2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ inline float2 ld_stream(const float2* p) {
    if (NT && (FFT_NT & 1)) { const f2v v = __builtin_nontemporal_load(reinterpret_cast<const f2v*>(p)); return make_float2(v.x, v.y); }
    return *p;
}
template <bool NT>
__device__ inline void st_stream(float2* p, float2 v) {
    if (NT && (FFT_NT & 2)) __builtin_nontemporal_store(f2v{v.x, v.y}, reinterpret_cast<f2v*>(p));
    else *p = v;
}
typedef float cf2 __attribute__((ext_vector_type(2)));
__device__ inline cf2 to_cf(float2 a) { return cf2{a.x, a.y}; }
__device__ inline float2 from_cf(cf2 a) { return make_float2(a.x, a.y); }
// a * w = a.xx * (w.x, w.y) + a.yy * (-w.y, w.x)
__device__ inline cf2 cmul_cf(cf2 a, cf2 w) {
    return __builtin_elementwise_fma(a.yy, cf2{-w.y, w.x}, a.xx * w);
}
__device__ inline float2 cmul(float2 a, float2 w) { return from_cf(cmul_cf(to_cf(a), to_cf(w))); }

constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v / 2); }

// In-register forward FFT of R points, decimation in frequency.  On return X[k]
// sits in v[bitrev(k)].  Fully unrolled: every index and twiddle is a constant.
template <int R>
__device__ inline void fft_reg(float2 (&vv)[R]) {
    cf2 v[R];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = to_cf(vv[i]);
#pragma unroll
    for (int h = R / 2; h >= 1; h /= 2) {
#pragma unroll
        for (int blk = 0; blk < R; blk += 2 * h) {
#pragma unroll
            for (int j = 0; j < h; ++j) {
                const cf2 a = v[blk + j], b = v[blk + j + h];
                v[blk + j] = a + b;
                const cf2 d = a - b;
                const int t = j * (16 / h);              // W_{2h}^j = W_32^{j * 32 / (2h)}
                if (t == 0) v[blk + j + h] = d;
                else if (t == 8) v[blk + j + h] = d.yx * cf2{1.0f, -1.0f};        // * (-i)
                else v[blk + j + h] = cmul_cf(d, cf2{kCos32[t], -kSin32[t]});
            }
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) vv[i] = from_cf(v[i]);
}

// ------------------------------------------------------------ strided C2C pass
// data[b * batch_stride + k * elem_stride + c], k < N = R1*R2, c < ncols (contiguous).
// POWER = true is the last (x) pass of the 3D transform fused with FFTPower's shell
// binning: batch index = ky, column = kz, row = kx; instead of storing delta_k each
// thread adds w |delta_k|^2 of its modes into the workgroup's LDS shell table, which
// is written out as one row of `partial` ([workgroup][shell], summed by the
// shell_partials_stage1/2 kernels in a fixed order).  The spectrum never goes back to HBM.
__device__ inline int tile_isqrt(int v) {
    // floor(sqrt(v)) for 0 <= v <= 3 * 512^2 from ONE raw v_sqrt_f32 (1 ulp) of v + 1/2: v and v + 1/2 are
    // exact in float; for a perfect square r^2 the argument's root is r + 1/(4r), for the largest v below
    // the next square it is (r+1) - 1/(4(r+1)) - both more than 4 ulp away from the integer at r = 887, more
    // below - so truncation gives r without the two integer repairs (8 instructions per mode in an
    // epilogue that is half of this pass's vector work).  Checked exhaustively by test_gpu_fft_tile.
    return (int)__builtin_amdgcn_sqrtf((float)v + 0.5f);
}

// test hook: out[v] = tile_isqrt(v)
__global__ void tile_isqrt_table_kernel(int* out, int count) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < count; v += gridDim.x * blockDim.x) out[v] = tile_isqrt(v);
}

// INV: the inverse transform (sum_k X_k e^{+2 pi i k n / N}) as conj(FFT(conj(X))): conjugate on load and on store.
// With INV the loads may come from a different array `src` of the same layout (data itself when null) and be
// masked to a shell: rows are k_x, the batch index k_y, the column k_z, and only modes with lo2 <= |m|^2 < hi2 pass
// (hi2 = 0: no mask) - the shell filter of the bispectrum estimator fused into its first inverse pass.
// PRUNING (hi2 > 0): every mode outside the sphere |m| < m_hi is zero, so
//   pass 0 (x; rows k_x, batch k_y, columns k_z) skips the tiles with k_y^2 + k_z0^2 >= hi2 entirely - no load, no
//          store: nobody reads them again;
//   pass 1 (y; rows k_y, batch x) skips the tiles with k_z0 >= m_hi and, inside the others, does not load the rows
//          with k_y^2 + k_z0^2 >= hi2 (the very tiles pass 0 left unwritten: both passes cut k_z into the same tiles);
//   the z pass (rows_c2r_kernel) reads k_z < m_hi only.
// A shell of radius N/4 moves 40 % of the full transform's bytes.
struct ShellMask { const float2* src; long long lo2, hi2; int ky0 = 0; int pass = 0;       // ky0 (POWER): global k_y index of batch 0 (slab blocks)
                   size_t src_elem_stride = 0, src_batch_stride = 0; };                    // strides of `src` when they differ from the destination's (0: the same)

// PACK: the stores go to `pack.out` in the layout the slab transpose sends (what ast_slab_pack makes of the array): row
// k_y of batch (plane) b lands in part k_y / c1 at ((part * nbatch + b) * c1 + k_y % c1) * pitch + column.
// Part `self_part` (the rank's own piece, which never travels) goes to self_out instead, as (nbatch, c1, pitch).
// SHELL BATCH (inverse passes of the bispectrum): one launch transforms up to SHELL_BATCH shells, blockIdx.y = shell - its own
// work array and radii.  Thirty-one shells used to be 93 launches of 0.06-0.4 ms each, most of them far too small to fill
// the chip to the end; batched, the shells of a launch fill each other's tails.
constexpr int SHELL_BATCH = 8;
#ifndef C2C_XCD_MODE
#define C2C_XCD_MODE 2              // batch rows -> XCDs in strided_c2c_kernel: 0 never, 1 inverse passes, 2 all passes
#endif
struct ShellBatch { int count = 0; float2* work[SHELL_BATCH]; long long lo2[SHELL_BATCH], hi2[SHELL_BATCH]; };
struct C2RBatch { int count = 0; const float2* in[SHELL_BATCH]; float* out[SHELL_BATCH]; int kmax[SHELL_BATCH]; };

// DISC layout (PACK = 2 stores, POWER loads): what is left of the half spectrum after the k_y pass once everything FFTPower
// drops is cut away, laid out for the slab transpose and for the last pass.  The k_y rows are dealt to `parts` owners in BLOCKS
// of R1 rows (the rows a workgroup of the k_y pass stores with one value of k2), balanced by the area of the Nyquist disc they
// cover; a part's plane is tile-major - for every 16-column k_z tile the part's rows that reach into the disc, in row order,
// 16 complex each - so the k_y pass stores whole 4-KB runs (R1 rows x 128 B) and every 128-byte piece sits on one line.
// One table entry per (k_z tile, row block): where sub-row 0 of the block would land in its part's plane (offb, in complex
// elements; sub-rows lo <= sub < hi exist), the part's plane size S, the sum of the lower parts' plane sizes, the part.
struct DiscEntry { int offb; unsigned S, cumS, lohi; };                   // lohi = lo | hi << 8 | part << 16
struct PackDst { float2* out = nullptr; unsigned c1_log2 = 0; unsigned nbatch = 0; float2* self_out = nullptr; unsigned self_part = ~0u; unsigned pitch = 0;
                 const DiscEntry* disc = nullptr; const unsigned short* gk = nullptr; unsigned part = 0;
                 unsigned xmajor = 0; };     // xmajor = N (one part only): element (x, row, col) of a tile at (row_pos * N + x) * 16 + col - the last pass reads N * 128 contiguous bytes

template <int R1, int R2, int C, bool POWER, bool INV = false, int PACK = 0>
__global__ void __launch_bounds__(C * (R1 > R2 ? R1 : R2))
// N = 1024 with binning: 128 VGPRs, so that two 8-wave workgroups fit a CU (see SPLIT below)
__attribute__((amdgpu_waves_per_eu((POWER || C > 16) && R1 * R2 >= 1024 ? 4 : 1, (POWER || C > 16) && R1 * R2 >= 1024 ? 4 : 8)))
strided_c2c_kernel(float2* __restrict__ data, const float2* __restrict__ tw_g, size_t elem_stride,
                   size_t ncols, size_t batch_stride, unsigned tiles_per_batch, float scale,
                   double* __restrict__ partial, const unsigned* __restrict__ edge_fall, ShellMask mask = ShellMask{nullptr, 0, 0},
                   PackDst pack = PackDst{}, ShellBatch sb = ShellBatch{}) {
    constexpr int N = R1 * R2;
    constexpr int NT = C * (R1 > R2 ? R1 : R2);
    constexpr int NB = N / 2 - 1;    // shells when POWER
    // SPLIT (N = 1024, binning pass): the stage-1 -> stage-2 exchange goes through LDS in two rounds, so the
    // buffer is 64 KB instead of 128 KB and TWO workgroups fit a CU: with one, the loads, the two register
    // FFTs and the binning of a tile run strictly one after the other (x pass 1.83 -> 1.54 ms).  The y pass
    // is bound by its 128-byte strided reads AND writes and loses a little with it (1.96 -> 2.03): not split.
    constexpr bool SPLIT = (POWER || C > 16) && R1 == R2 && N * C * sizeof(float2) > 64 * 1024;
    constexpr int YN = SPLIT ? N / 2 : N;
    extern __shared__ float2 lds[];
    float2* Y = lds;                 // [n2][k1][c]  (SPLIT: [n2 mod R2/2][k1][c])
    float2* tw = lds + YN * C;       // exp(-2 pi i m / N)
    double* shell = reinterpret_cast<double*>(lds + YN * C + N);    // [NB + 1] when POWER
    unsigned tile = blockIdx.x % tiles_per_batch, b = blockIdx.x / tiles_per_batch;
#if C2C_XCD_MODE
    // A 16-column tile row is 128 bytes at an 8-byte-aligned address (row pitch N / 2 + 1): every line it touches is shared
    // with the neighbouring column tile.  Consecutive workgroups go to consecutive XCDs (observed round-robin placement;
    // a matter of speed only), so neighbouring tiles sat behind different L2s and each fetched the shared lines - the
    // masked inverse passes read 1.94 x their bytes (scripts/pmc_per_launch.py).  Here the tiles of one batch row follow each
    // other on ONE XCD: batch b runs on XCD b mod 8.  512^3 bispectrum: x / y inverse passes 6.9 -> 5.9 ms; forward passes
    // 512^3 0.52 -> 0.46 ms, 256^3 0.069 -> 0.050 ms, 1024^3 (32-column tiles) -0.05 ms per step.
    if ((INV || C2C_XCD_MODE == 2) && (gridDim.x / tiles_per_batch) % 8u == 0u) {
        const unsigned xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
        b = (slot / tiles_per_batch) * 8u + xcd;
        tile = slot % tiles_per_batch;
    }
#endif
    const size_t c0 = (size_t)tile * C;
    if (INV && sb.count > 0) {                        // this workgroup's shell of the batch (uniform)
        data = sb.work[blockIdx.y];
        mask.lo2 = sb.lo2[blockIdx.y];
        mask.hi2 = sb.hi2[blockIdx.y];
    }
    if (INV && mask.hi2 > 0) {                        // pruning: block-uniform exits before any barrier
        const long long kb = (long long)((int)b > N / 2 ? (int)b - N : (int)b);
        const long long r2 = (long long)(c0 * c0) + (mask.pass == 0 ? kb * kb : 0);
        if (r2 >= mask.hi2) return;
    }
    // FORWARD PRUNING (mask.hi2 = (N/2)^2 on the two strided passes of the power pipeline): FFTPower keeps |m| < N/2 only
    // (power_spectrum_3d.py:189-195; a vector of norm exactly N/2 may fall into the last shell under the float64 rule, so the
    // edge itself stays).  After the y pass a row (k_y, 16-column tile from k_z0) with k_y^2 + k_z0^2 > (N/2)^2 holds no mode
    // any shell takes, whatever k_x: the y pass does not store it (below), and the binning pass leaves that tile out - 21.5 %
    // of the half plane, neither written nor read again.  The skipped workgroup still owns a row of `partial`: zeros.
    // DISC source (binning pass over a part's block in the disc layout): batch b is the part's local row, block b / R1 of the
    // part is row block gk of the lattice; the table says where the tile's rows of that block start - or that this row
    // lies outside the disc for this tile: nothing stored, nothing to bin
    const float2* disc_base = nullptr;
    unsigned disc_S = 0;
    int disc_ky = 0;
    if (!INV && POWER && pack.disc != nullptr) {
        const unsigned subb = b % R1, gk = pack.gk[pack.part * (pack.nbatch / R1) + b / R1];
        const DiscEntry e = pack.disc[(size_t)tile * R2 + gk];
        const unsigned lo = e.lohi & 255u, hi = (e.lohi >> 8) & 255u;
        if (subb - lo >= hi - lo) {
            for (int i = threadIdx.x; i < NB; i += NT) partial[((size_t)b * tiles_per_batch + tile) * NB + i] = 0.0;
            return;
        }
        disc_ky = (int)(gk * R1 + subb);
        if (pack.xmajor) {                            // x-major tiles: the N planes' pieces of this (row, tile) are contiguous
            disc_base = data + ((long long)e.offb + (long long)(subb * 16u)) * (long long)pack.xmajor;
            disc_S = 16u;
        } else {
            disc_base = data + ((long long)e.offb + (long long)(subb * 16u));
            disc_S = e.S;
        }
    } else if (!INV && POWER && mask.hi2 > 0) {
        const long long kyi = (long long)b + mask.ky0, ky = kyi > N / 2 ? kyi - N : kyi;
        if (ky * ky + (long long)(c0 * c0) > mask.hi2) {
            for (int i = threadIdx.x; i < NB; i += NT) partial[((size_t)b * tiles_per_batch + tile) * NB + i] = 0.0;
            return;
        }
    }
    for (int i = threadIdx.x; i < N; i += NT) tw[i] = tw_g[i];
    if (POWER)
        for (int i = threadIdx.x; i <= NB; i += NT) shell[i] = 0.0;
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const bool col_ok = (unsigned)c0 + (unsigned)c < (unsigned)ncols;       // 32-bit: the host checks ncols < 2^31
    // AST_BIN_FLOAT64: bit k2 of this thread's word says that its epilogue mode (row sub + R1 * k2, column c) has an
    // integer norm AND nbodykit's float64 comparison puts it one shell lower.  The words are data independent and
    // come precomputed (edge_fall_kernel): one register, no double-precision code in this register-tight kernel.
    float2 u[R2];
    {                                                 // stage 1: task (c, n2 = sub)
        const bool task1 = sub < R2;
        float2 v[R1];
        // unconditional loads (columns past the end re-read the last valid one, idle sub-tasks the
        // last row block): predicated loads compile to a branch each
        // address = UNIFORM base (batch and row block: scalar registers) + one 32-bit per-lane offset shared by all
        // R1 loads (global_load ... v_off, s[base]): per-load 64-bit lane addresses cost two VGPRs each while in flight
        // (the masked first inverse pass may read a source of another row pitch than the array it writes)
        const bool own = INV && mask.src != nullptr && mask.src_elem_stride != 0;
        size_t es_in = own ? mask.src_elem_stride : elem_stride;
        const float2* ubase = (INV && mask.src ? mask.src : data) + (size_t)b * (own ? mask.src_batch_stride : batch_stride);
        const int lsub = task1 ? sub : R2 - 1;
        uint32_t voff = (uint32_t)lsub * (uint32_t)es_in + min((uint32_t)c0 + (uint32_t)c, (uint32_t)ncols - 1u);     // host checks: < 2^29
        if (POWER && !INV && disc_base != nullptr) {      // rows = planes, S apart; the tile's 16 columns are contiguous
            es_in = disc_S;
            ubase = disc_base;
            voff = (uint32_t)lsub * disc_S + (uint32_t)c;
        }
        if (INV && mask.hi2 > 0 && mask.pass == 1) {
            const long long c02 = (long long)(c0 * c0);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {         // rows of tiles pass 0 skipped: zero, and not in memory
                const int row = n1 * R2 + lsub;
                const long long ky = row > N / 2 ? row - N : row;
                v[n1] = make_float2(0.f, 0.f);
                if (ky * ky + c02 < mask.hi2) v[n1] = (ubase + (size_t)(n1 * R2) * es_in)[voff];
            }
        } else if (INV && mask.hi2 > 0) {
            // pass 0: the R2 rows n1 R2 .. n1 R2 + R2 - 1 (one per row of threads) lie outside the sphere together when the
            // smallest |k_x| among them does - a test on scalars, the same for the whole workgroup: no load (the x pass read
            // all N rows of every column inside the disc before, 5.2 GB for 3.4 GB of nonzero rows over the 31 shells)
            const long long kyb = (long long)((int)b > N / 2 ? (int)b - N : (int)b);
            const long long r2yz = kyb * kyb + (long long)(c0 * c0);
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) {
                const int lo = n1 * R2, hi = lo + R2 - 1;
                const long long kxmin = hi <= N / 2 ? lo : (lo >= N / 2 ? N - hi : 0);
                v[n1] = make_float2(0.f, 0.f);
                if (kxmin * kxmin + r2yz < mask.hi2) v[n1] = (ubase + (size_t)(n1 * R2) * es_in)[voff];
            }
        } else {
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1] = ld_stream<(!INV && N >= 1024)>(ubase + (size_t)(n1 * R2) * es_in + voff);
        }
        if (INV) {
            if (mask.hi2 > 0 && mask.pass == 0) {
                const long long kz = (long long)min(c0 + c, ncols - 1), ky = (long long)((int)b > N / 2 ? (int)b - N : (int)b);
                const long long m2yz = ky * ky + kz * kz;
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) {
                    const int row = n1 * R2 + lsub;
                    const long long kx = row > N / 2 ? row - N : row, m2 = kx * kx + m2yz;
                    if (m2 < mask.lo2 || m2 >= mask.hi2) v[n1] = make_float2(0.f, 0.f);
                }
            }
#pragma unroll
            for (int n1 = 0; n1 < R1; ++n1) v[n1].y = -v[n1].y;
        }
        fft_reg<R1>(v);
        __syncthreads();                              // twiddle table is in LDS
        if (!SPLIT) {
            if (task1) {
#pragma unroll
                for (int k1 = 0; k1 < R1; ++k1) {
                    float2 y = v[bitrev(k1, ilog2(R1))];
                    if (k1 != 0) y = cmul(y, tw[sub * k1]);
                    Y[(sub * R1 + k1) * C + c] = y;
                }
            }
            __syncthreads();
            if (sub < R1) {                           // stage 2: task (c, k1 = sub)
#pragma unroll
                for (int n2 = 0; n2 < R2; ++n2) u[n2] = Y[(n2 * R1 + sub) * C + c];
            }
        } else {
            // R1 == R2: every thread has a task in both stages.  Round r carries the n2 of half r:
            // its producers write all their k1 (and are done with v), every thread collects 16 of
            // its 32 inputs.  Peak registers: v plus half of u.
            constexpr int H = R2 / 2;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (sub / H == half) {                // wave-uniform: C = 16, H = 16 -> 4 waves per half
#pragma unroll
                    for (int k1 = 0; k1 < R1; ++k1) {
                        float2 y = v[bitrev(k1, ilog2(R1))];
                        if (k1 != 0) y = cmul(y, tw[sub * k1]);
                        Y[((sub - half * H) * R1 + k1) * C + c] = y;
                    }
                }
                __syncthreads();
#pragma unroll
                for (int nh = 0; nh < H; ++nh) u[half * H + nh] = Y[(nh * R1 + sub) * C + c];
                if (half == 0) __syncthreads();       // round 0 has been read: round 1 may overwrite it
            }
        }
    }
    if (sub < R1) {
        fft_reg<R2>(u);
        // (store addresses and the column test rebuilt from the thread id here: the 32-column variant is held to 128 VGPRs
        // and would otherwise carry - spill - them across the two register FFTs)
        int tid_s = (int)threadIdx.x;
        if (C > 16) asm volatile("" : "+v"(tid_s));
        const int cs = C > 16 ? tid_s % C : c;
        const bool ok_s = C > 16 ? (unsigned)c0 + (unsigned)cs < (unsigned)ncols : col_ok;
        if (PACK == 2 && !POWER) {
            // DISC stores: per k2 one table entry (scalar loads), a uniform base and the thread id as the lane offset - the
            // R1 x 16 values of a row block land in one contiguous run.  Columns past ncols hold copies of the last one.
            static_assert(PACK != 2 || C == 16, "the disc layout's tiles are 16 columns");
            // the tile's R2 entries in ONE vector load - lane l holds entry l mod R2 - and v_readlane per k2 (32 scalar loads,
            // each waited for in turn, cost the pass a quarter of its time)
            const int4 ev = reinterpret_cast<const int4*>(pack.disc + (size_t)tile * R2)[threadIdx.x % R2];
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                float2 x = u[bitrev(k2, ilog2(R2))];
                x.x *= scale;
                x.y *= scale;
                const int offb = __builtin_amdgcn_readlane(ev.x, k2);
                const unsigned S = (unsigned)__builtin_amdgcn_readlane(ev.y, k2), cumS = (unsigned)__builtin_amdgcn_readlane(ev.z, k2);
                const unsigned lohi = (unsigned)__builtin_amdgcn_readlane(ev.w, k2);
                const unsigned lo = lohi & 255u, hi = (lohi >> 8) & 255u, part = lohi >> 16;
                if (pack.xmajor) {
                    float2* const ub = pack.self_out + ((long long)offb * (long long)pack.xmajor + (long long)(b * 16u));
                    if ((unsigned)sub - lo < hi - lo)
                        st_stream<(N >= 1024)>(ub + ((uint32_t)sub * 16u * pack.xmajor + (uint32_t)(threadIdx.x & 15u)), x);
                    continue;
                }
                float2* const ub = (part == pack.self_part ? pack.self_out : pack.out + (size_t)pack.nbatch * cumS)
                                   + ((size_t)b * S + (long long)offb);
                if ((unsigned)sub - lo < hi - lo) st_stream<(N >= 1024)>(ub + threadIdx.x, x);
            }
        } else if (ok_s && !POWER) {
            float2* const base = data + (size_t)b * batch_stride + c0 + cs;
            // forward pruning: rows are k_y, the tile starts at k_z0 = c0; keep k_y^2 <= room.  Decided per WAVE on scalars (the
            // wave's smallest |k_y| of the row block; a wave holds 64 / C consecutive `sub`): a per-lane test cost 16 VGPRs
            // in a kernel held to 128, and the one to three rows a wave stores beyond the disc are never read.
            const int room = (!INV && mask.hi2 > 0) ? (int)mask.hi2 - (int)(c0 * c0) : 0x7fffffff;
            const int wsub0 = __builtin_amdgcn_readfirstlane((int)threadIdx.x) / C, wsub1 = wsub0 + 64 / C - 1;
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                float2 x = u[bitrev(k2, ilog2(R2))];
                x.x *= scale;
                x.y *= INV ? -scale : scale;
                if (!INV) {
                    const int aky = R1 * k2 >= N / 2 ? N - (wsub1 + R1 * k2) : wsub0 + R1 * k2;    // smallest |k_y| in the wave
                    if (aky * aky > room) continue;
                }
                if (PACK) {
                    // c1 = rows per part >= R1 (host-checked): the part of row sub + R1 k2 and its row block inside the
                    // part do not depend on the lane - a uniform base per k2 plus ONE 32-bit lane offset for all stores
                    const unsigned part = (unsigned)(R1 * k2) >> pack.c1_log2, r0 = (unsigned)(R1 * k2) & ((1u << pack.c1_log2) - 1u);
                    float2* const ub = part == pack.self_part
                        ? pack.self_out + ((((size_t)b) << pack.c1_log2) + r0) * pack.pitch
                        : pack.out + ((((size_t)part * pack.nbatch + b) << pack.c1_log2) + r0) * pack.pitch;
                    st_stream<(N >= 1024)>(ub + ((uint32_t)sub * pack.pitch + (uint32_t)c0 + (uint32_t)cs), x);
                } else {
                    st_stream<(!INV && N >= 1024)>(base + (size_t)(sub + R1 * k2) * elem_stride, x);
                }
            }
        }
    }
    int tid_e = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_e));                 // the epilogue's column comes from the thread id again: not carried (spilled) from the top
    if (POWER && (unsigned)c0 + (unsigned)(tid_e % C) < (unsigned)ncols && sub < R1) {
        // one ds_add_f64 per mode; merging a wave's equal-shell lanes first (ballot + cross-lane
        // adds) measured slower: 16 kz x 4 kx lanes rarely share one shell
        // (the word is fetched HERE, not at the top: nothing of this epilogue may stay live across the two register
        // FFTs - the kernel is held to 128 VGPRs and every long-lived value there turned into scratch traffic)
        unsigned fallmask = 0;
        const int kyi = disc_base != nullptr ? disc_ky : (int)b + mask.ky0;
        if (edge_fall) fallmask = edge_fall[((size_t)(disc_base != nullptr ? (unsigned)disc_ky : b) * tiles_per_batch + tile) * NT + threadIdx.x];
        const int kz = (int)c0 + tid_e % C;
        const int ky = kyi > N / 2 ? kyi - N : kyi;
        const int m2yz = ky * ky + kz * kz;
        const float w = (kz > 0 && kz < N / 2) ? 2.0f : 1.0f;
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) {
            const float2 x = u[bitrev(k2, ilog2(R2))];
            const int row = sub + R1 * k2;
            const int kx = row > N / 2 ? row - N : row;
            int sh = tile_isqrt(kx * kx + m2yz);                        // shell = sh - 1; 0 is DC
            // lattice vectors of integer norm sit on a shell edge: nbodykit's float64 comparison decides (rare path)
            sh -= (int)((fallmask >> k2) & 1u);
            // |delta_k|^2 in fp32 (one rounding of 6e-8 per mode, random over the >= 18 modes of a
            // shell), accumulated in double
            if (sh >= 1 && sh <= NB) atomicAdd(&shell[sh], (double)(fmaf(x.x, x.x, x.y * x.y) * w));
        }
    }
    if (POWER) {
        __syncthreads();
        const double s2 = (double)scale * (double)scale;
        for (int i = threadIdx.x; i < NB; i += NT) partial[((size_t)b * tiles_per_batch + tile) * NB + i] = shell[i + 1] * s2;
    }
}

// psum[bin] += pnorm * sum over workgroups of partial[wg][bin], fixed summation order, in two
// stages that both read whole rows (lanes = consecutive bins): REDUCE_ROWS blocks each add up a
// contiguous range of workgroup rows, then one block adds those.  (One block per bin striding down
// the matrix read 8 of every 4096 bytes it touched: 0.15 ms for 138 MB at 1024^3.)
constexpr int REDUCE_ROWS = 1024;
__global__ void __launch_bounds__(256)
shell_partials_stage1_kernel(const double* __restrict__ partial, size_t nwg, int nb, double* __restrict__ partial2) {
    const size_t per = (nwg + REDUCE_ROWS - 1) / REDUCE_ROWS;
    const size_t g0 = (size_t)blockIdx.x * per, g1 = g0 + per < nwg ? g0 + per : nwg;
    for (int bin = threadIdx.x; bin < nb; bin += 256) {
        double acc = 0.0;
#pragma unroll 8
        for (size_t g = g0; g < g1; ++g) acc += partial[g * nb + bin];
        partial2[(size_t)blockIdx.x * nb + bin] = acc;
    }
}
// 8 bins per block: 32 row groups x 8 bins, each thread adds REDUCE_ROWS / 32 rows, then the groups in order
__global__ void __launch_bounds__(256)
shell_partials_stage2_kernel(const double* __restrict__ partial2, int nb, double pnorm, int first_bin,
                             double* __restrict__ psum) {
    __shared__ double part[32][8];
    const int b8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int bin = blockIdx.x * 8 + b8;
    double acc = 0.0;
    if (bin < nb) {
#pragma unroll 8
        for (int r = grp; r < REDUCE_ROWS; r += 32) acc += partial2[(size_t)r * nb + bin];
    }
    part[grp][b8] = acc;
    __syncthreads();
    if (grp == 0 && bin < nb && bin >= first_bin) {       // bins below first_bin come from the low-k channel
        double t = 0.0;
        for (int k = 0; k < 32; ++k) t += part[k][b8];
        psum[bin] += pnorm * t;
    }
}

// ------------------------------------------------------- contiguous-row R2C pass
// in: nrows rows of N = 2*M reals (pitch in_pitch reals); out: rows of M+1 complex.
// (constants of the low-k channel, described further down; the z pass below can produce its z sums on the side)
constexpr int MLOW_FWD = 5, MBOX_FWD = MLOW_FWD + 1;
__device__ constexpr double kC16f[16] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                                         -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                         -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                         0.38268343236508977, 0.70710678118654752, 0.92387953251128674};
__device__ constexpr double kS16f[16] = {0.0, -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                         -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                         0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1.0,
                                         0.92387953251128674, 0.70710678118654752, 0.38268343236508977};

// LOWK: the thread's 2 R1 samples x[2 R2 n1 + 2 n2 + p] are in registers right after the load, so the low-k channel's
// z sums S_kz = sum_z x[z] e^{-2 pi i kz z / N}, kz <= 6, are formed here in double instead of by a second pass over
// the grid (lowk_z_kernel: 4.3 GB, 1.2 ms at 1024^3): per p the R1-term sum over n1 with the R1-th roots of unity as
// constants (e^{-2 pi i 2 R2 / N} = e^{-2 pi i / R1}), then T_0 A + T_1 B with the lane's factors A = w^{2 kz n2},
// B = w^{kz (2 n2 + 1)} (table lowk_lane, double), then a transposed reduction over the row's R2 lanes.  Output as
// lowk_z_kernel's.
// M = R1*R2.  One workgroup transforms C rows.
// FOLDW = 2 / 3 (CIC / TSC): `in` is the grid a deferred-fold paint left (AST_PAINT_DEFER_FOLD) and `rec`
// its halo records; the up to three record lines that end in a border row are added as the row is
// loaded — sum of the records first, then onto the row, the order of column_fold_kernel, so the
// result is bit-identical to folding first.  Rows are (x, y) lines of the periodic n^3 grid.
// SlabFold (FOLDW != 0, slab buffers): the rows are the (x, y) lines of the buffer planes xb0, xb0 + 1, ... of a slab buffer
// of ntx tile rows (not periodic in x); only planes of the tile rows [row_lo, row_hi) are folded on load - the rows that hold
// ghost planes were folded by the paint's own FOLD stage before their planes went to the neighbours.  ntx = 0: the whole
// periodic grid, every row folded.
struct SlabFold { int xb0 = 0, ntx = 0, row_lo = 0, row_hi = 0; bool ranges = false; };      // ranges: plane ranges of a buffer (ntx = 0: the periodic grid)

template <int R1, int R2, int C, int FOLDW, bool LOWK = false>
__global__ void __launch_bounds__(C * R2)
__attribute__((amdgpu_waves_per_eu(LOWK ? 4 : 1, LOWK ? 4 : 8)))     // LOWK: keep two workgroups per CU (128 VGPRs)
rows_r2c_kernel(const float* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw_g,
                size_t nrows, size_t in_pitch, size_t out_pitch, float scale, float mean,
                const float* __restrict__ rec, const double2* __restrict__ lowk_lane = nullptr,
                double* __restrict__ lowz = nullptr, SlabFold sf = SlabFold{}) {
    constexpr int M = R1 * R2, N = 2 * M;
    constexpr int NT = C * R2;
    constexpr int R2P = R2 + 1;
    constexpr int MP = M + 1;
    extern __shared__ float2 lds[];
    float2* Y = lds;                     // stage buffer [r][k1][n2] (padded), then Z[r][k]
    float2* tw = lds + C * (R1 * R2P > MP ? R1 * R2P : MP);     // exp(-2 pi i m / N), m < N
    for (int i = threadIdx.x; i < N; i += NT) tw[i] = tw_g[i];

    const size_t row0 = (size_t)blockIdx.x * C;
    const int n2 = threadIdx.x % R2, r = threadIdx.x / R2;        // stage-1 task (r, n2)
    {
        // z[j] = x[2j] + i x[2j+1]; rows past the end re-read the last one (unconditional loads)
        const float2* zin = reinterpret_cast<const float2*>(in + min(row0 + r, nrows - 1) * in_pitch);
        float2 v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) v[n1] = ld_stream<(N >= 1024)>(zin + n1 * R2 + n2);
        if (FOLDW != 0) {
            constexpr int W = FOLDW != 0 ? FOLDW : 2;
            const int ng = (int)(2 * M);
            const size_t row = min(row0 + r, nrows - 1);
            const float* src[3];
            int ns;
            if (sf.ntx == 0) {
                const int xb = sf.xb0 + (int)(row / ng), trow = xb / ast::TX;
                ns = !sf.ranges || (trow >= sf.row_lo && trow < sf.row_hi)
                         ? ast::halo_sources<float, W>(rec, xb, (int)(row % ng), ng, ng / ast::TX, ng / ast::TY, src) : 0;
            } else {
                const int xb = sf.xb0 + (int)(row / ng), trow = xb / ast::TX;
                ns = trow >= sf.row_lo && trow < sf.row_hi
                         ? ast::halo_sources<float, W>(rec, xb, (int)(row % ng), ng, sf.ntx, ng / ast::TY, src, false) : 0;
            }
            if (ns > 0) {                    // 15 of 64 rows (CIC); the loads and adds stay inside the branch
                float2 h[R1];
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) h[n1] = reinterpret_cast<const float2*>(src[0])[n1 * R2 + n2];
                if (ns > 1) {
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const float2 t = reinterpret_cast<const float2*>(src[1])[n1 * R2 + n2];
                        h[n1].x += t.x;
                        h[n1].y += t.y;
                    }
                }
                if (ns > 2) {
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const float2 t = reinterpret_cast<const float2*>(src[2])[n1 * R2 + n2];
                        h[n1].x += t.x;
                        h[n1].y += t.y;
                    }
                }
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) {
                    v[n1].x += h[n1].x;
                    v[n1].y += h[n1].y;
                }
            }
        }
        if (LOWK) {
            static_assert(!LOWK || ((R1 == 16 || R1 == 8) && (R2 == 32 || R2 == 16) && MBOX_FWD == 6), "R1-th roots from the 16th-root table; 32 or 16 lanes per row");
            double acc[14];
#pragma unroll
            for (int e = 0; e < 14; ++e) acc[e] = 0.0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {                    // one parity at a time: 16 doubles live, not 32
                double f[R1];
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) f[n1] = (double)(p ? v[n1].y : v[n1].x);
#pragma unroll
                for (int kz = 0; kz <= MBOX_FWD; ++kz) {
                    double tr = 0.0, ti = 0.0;
#pragma unroll
                    for (int n1 = 0; n1 < R1; ++n1) {
                        const int rr = ((kz * n1) % R1) * (16 / R1);     // e^{-2 pi i kz n1 / R1}: a compile-time constant
                        if (rr == 0) tr += f[n1];
                        else if (rr == 4) ti -= f[n1];
                        else if (rr == 8) tr -= f[n1];
                        else if (rr == 12) ti += f[n1];
                        else { tr = fma(f[n1], kC16f[rr], tr); ti = fma(f[n1], kS16f[rr], ti); }
                    }
                    const double2 a = lowk_lane[(n2 * (MBOX_FWD + 1) + kz) * 2 + p];
                    acc[2 * kz] = fma(a.x, tr, fma(-a.y, ti, acc[2 * kz]));
                    acc[2 * kz + 1] = fma(a.x, ti, fma(a.y, tr, acc[2 * kz + 1]));
                }
            }
            // 14 sums over the row's R2 lanes (a wave holds 64 / R2 rows): 14 -> 7 (+1 zero) -> 4 -> 2 -> 1, with 32 lanes
            // then the last pair
            const int lane = threadIdx.x & 63;
            auto level = [&](auto count_tag, int mask) {
                constexpr int COUNT = decltype(count_tag)::value;
                const bool upper = (lane & mask) != 0;
#pragma unroll
                for (int i = 0; i < COUNT / 2; ++i) {
                    const double keep = upper ? acc[i + COUNT / 2] : acc[i];
                    const double send = upper ? acc[i] : acc[i + COUNT / 2];
                    acc[i] = keep + __shfl_xor(send, mask, 64);
                }
            };
            constexpr int TOP = R2 / 2;                      // 16 or 8
            level(std::integral_constant<int, 14>{}, TOP);
            acc[7] = 0.0;
            level(std::integral_constant<int, 8>{}, TOP / 2);
            level(std::integral_constant<int, 4>{}, TOP / 4);
            level(std::integral_constant<int, 2>{}, TOP / 8);
            if (R2 == 32) acc[0] += __shfl_xor(acc[0], 1, 64);
            constexpr int SH = R2 == 32 ? 1 : 0;             // the lane bits below the four that select the element
            const int sel = lane >> SH;
            const int sub = ((sel >> 2) & 1) * 4 + ((sel >> 1) & 1) * 2 + (sel & 1);
            if ((R2 == 16 || (lane & 1) == 0) && sub < 7 && row0 + r < nrows)
                lowz[(row0 + r) * (size_t)(2 * (MBOX_FWD + 1)) + ((sel >> 3) & 1) * 7 + sub] = acc[0];
        }
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            v[n1].x -= mean;                 // only the (discarded) DC mode sees the offset; fp32 round-off
            v[n1].y -= mean;                 // of every other mode no longer scales with it
        }
        fft_reg<R1>(v);
        __syncthreads();
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            float2 y = v[bitrev(k1, ilog2(R1))];
            if (k1 != 0) y = cmul(y, tw[2 * n2 * k1]);            // W_M = W_N^2
            Y[(r * R1 + k1) * R2P + n2] = y;
        }
    }
    __syncthreads();
    float2 u[R2];
    const int k1 = threadIdx.x % R1, r2 = threadIdx.x / R1;       // stage-2 task (r2, k1): C*R1 of the C*R2 threads
    const bool task2 = r2 < C;
    if (task2) {
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = Y[(r2 * R1 + k1) * R2P + j];
        fft_reg<R2>(u);
    }
    __syncthreads();                                              // everyone has read Y: reuse it as Z[r][k]
    if (task2) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) Y[r2 * MP + k1 + R1 * k2] = u[bitrev(k2, ilog2(R2))];
    }
    __syncthreads();
    // even/odd split: X[k] = (Zk + conj(Zm))/2 - i w^k (Zk - conj(Zm))/2, m = M - k, w = e^{-2 pi i / N}
    for (int i = threadIdx.x; i < C * (M / 2 + 1); i += NT) {
        const int rr = i / (M / 2 + 1), k = i % (M / 2 + 1);
        if (row0 + rr >= nrows) continue;
        const float2 zk = Y[rr * MP + k];
        const float2 zm = Y[rr * MP + ((M - k) & (M - 1))];
        float2* orow = out + (row0 + rr) * out_pitch;
        if (k == 0) {
            orow[0] = make_float2((zk.x + zk.y) * scale, 0.f);
            orow[M] = make_float2((zk.x - zk.y) * scale, 0.f);
            continue;
        }
        const float2 e = make_float2(0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y));      // (Zk + conj Zm)/2
        const float2 o = make_float2(0.5f * (zk.x - zm.x), 0.5f * (zk.y + zm.y));      // (Zk - conj Zm)/2
        const float2 w = tw[k];
        const float2 t = cmul(o, w);                                                    // w^k * o
        // X[k] = e - i t ;  X[M-k] = conj(e) - i * conj(w^k)... = conj(e + i t)
        st_stream<(N >= 1024)>(orow + k, make_float2((e.x + t.y) * scale, (e.y - t.x) * scale));
        st_stream<(N >= 1024)>(orow + M - k, make_float2((e.x - t.y) * scale, (-e.y - t.x) * scale));
    }
}

// ------------------------------------------------------- contiguous-row C2R pass
// in: nrows rows of M + 1 complex (pitch in_pitch complex, Hermitian half spectrum of a real row, X[0] and X[M] real
// as far as they matter: their imaginary parts are ignored like in every C2R); out: rows of N = 2 M reals,
// x[n] = sum_{k=0}^{N-1} X[k] e^{+2 pi i k n / N} (unnormalised).  The inverse of rows_r2c_kernel:
//   Z[k] = (X[k] + conj X[M-k]) / 2  +  i w^{-k} (X[k] - conj X[M-k]) / 2,   w = e^{-2 pi i / N},  k < M
//   z = IFFT_M(Z) (as conj(FFT(conj Z)), the two-stage register FFT of the forward pass),  x[2j] + i x[2j+1] = 2 z[j].
template <int R1, int R2, int C>
__global__ void __launch_bounds__(C * R2)
rows_c2r_kernel(const float2* __restrict__ in, float* __restrict__ out, const float2* __restrict__ tw_g,
                size_t nrows, size_t in_pitch, size_t out_pitch, float scale, int kmax, C2RBatch cb = C2RBatch{}) {
    if (cb.count > 0) { in = cb.in[blockIdx.y]; out = cb.out[blockIdx.y]; kmax = cb.kmax[blockIdx.y]; }      // this workgroup's shell
    constexpr int M = R1 * R2, N = 2 * M;
    constexpr int NT = C * R2;
    constexpr int R2P = R2 + 1;
    constexpr int MP = M + 1;
    extern __shared__ float2 lds[];
    float2* Y = lds;                     // X rows [r][k <= M], then the stage buffer [r][k1][n2] (padded), then z[r][j]
    float2* tw = lds + C * (R1 * R2P > MP ? R1 * R2P : MP);     // exp(-2 pi i m / N), m < N
    for (int i = threadIdx.x; i < N; i += NT) tw[i] = tw_g[i];
    const size_t row0 = (size_t)blockIdx.x * C;
    {
        // R2 threads per row, k = q + R2 i.  All loads of a thread are issued before the first LDS store (a load
        // inside a per-lane condition is a branch and a full wait each); k >= kmax is zero by construction and not
        // read - the cut is uniform over the workgroup, in steps of R2.
        const int rr = threadIdx.x / R2, q = threadIdx.x % R2;
        const float2* src = in + min(row0 + rr, nrows - 1) * in_pitch;
        float2 tmp[R1 + 1];
#pragma unroll
        for (int i = 0; i <= R1; ++i) {
            tmp[i] = make_float2(0.f, 0.f);
            if (i * R2 < kmax) tmp[i] = src[min(q + R2 * i, M)];
        }
#pragma unroll
        for (int i = 0; i <= R1; ++i) {
            const int k = q + R2 * i;
            if (k <= M) Y[rr * MP + k] = k < kmax ? tmp[i] : make_float2(0.f, 0.f);
        }
    }
    __syncthreads();
    const int n2 = threadIdx.x % R2, r = threadIdx.x / R2;        // stage-1 task (r, n2)
    float2 v[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
        const int k = n1 * R2 + n2;                               // < M
        const float2 xk = Y[r * MP + k], xm = Y[r * MP + M - k];
        const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));      // (X[k] + conj X[M-k]) / 2
        const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));      // (X[k] - conj X[M-k]) / 2
        const float2 w = tw[k];                                                        // w^k; conj = w^{-k}
        const float2 t = make_float2(d.x * w.x + d.y * w.y, d.y * w.x - d.x * w.y);    // w^{-k} d
        // Z = e + i t;  the FFT below runs on conj(Z)
        v[n1] = make_float2(e.x - t.y, -(e.y + t.x));
    }
    __syncthreads();                                              // everyone has read the X rows: Y is reused
    fft_reg<R1>(v);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
        float2 y = v[bitrev(k1, ilog2(R1))];
        if (k1 != 0) y = cmul(y, tw[2 * n2 * k1]);                // W_M = W_N^2
        Y[(r * R1 + k1) * R2P + n2] = y;
    }
    __syncthreads();
    float2 u[R2];
    const int k1 = threadIdx.x % R1, r2 = threadIdx.x / R1;       // stage-2 task (r2, k1)
    const bool task2 = r2 < C;
    if (task2) {
#pragma unroll
        for (int j = 0; j < R2; ++j) u[j] = Y[(r2 * R1 + k1) * R2P + j];
        fft_reg<R2>(u);
    }
    __syncthreads();
    if (task2) {
#pragma unroll
        for (int k2 = 0; k2 < R2; ++k2) Y[r2 * MP + k1 + R1 * k2] = u[bitrev(k2, ilog2(R2))];
    }
    __syncthreads();
    // z[j] = conj(result[j]); x[2j] = 2 Re z, x[2j+1] = 2 Im z
    const float s2 = 2.0f * scale;
#ifndef C2R_ST
#define C2R_ST 2                  // 0 = 8-byte stores, 1 = 16-byte stores, 2 = 16-byte NONTEMPORAL stores.  Measured on the 512^3 bispectrum
                                  // (31 shells): this pass 6.5 / 6.0 / 6.1 ms, and with 2 the NEXT shell's masked x / y passes 7.5 -> 6.8 ms:
                                  // the 0.5 GB real cube no longer pushes the 0.5 GB spectrum they re-read out of the Infinity Cache
#endif
    if (C2R_ST) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        for (int i = threadIdx.x; i < C * (M / 2); i += NT) {
            const int rr = i / (M / 2), j = (i % (M / 2)) * 2;
            if (row0 + rr >= nrows) continue;
            const float2 z0 = Y[rr * MP + j], z1 = Y[rr * MP + j + 1];
            const f4v o = {z0.x * s2, -z0.y * s2, z1.x * s2, -z1.y * s2};
            f4v* dst = reinterpret_cast<f4v*>(out + (row0 + rr) * out_pitch) + j / 2;
            if (C2R_ST == 2) __builtin_nontemporal_store(o, dst); else *dst = o;
        }
        return;
    }
    for (int i = threadIdx.x; i < C * M; i += NT) {
        const int rr = i / M, j = i % M;
        if (row0 + rr >= nrows) continue;
        const float2 z = Y[rr * MP + j];
        reinterpret_cast<float2*>(out + (row0 + rr) * out_pitch)[j] = make_float2(z.x * s2, -z.y * s2);
    }
}

// ------------------------------------------- z pass of ALL shells fused with the triangle sums (bispectrum)
// The estimator's last two steps used to be: per shell a z pass that WRITES the real cube (31 x 0.5 GB at 512^3), then one
// kernel that READS the 31 cubes back and forms the triangle sums - 33 of the 65 GB the whole bispectrum moves.  Here one
// workgroup takes ONE (x, y) row of EVERY shell: "row" r of the C2R transform below is shell r's row (its own scratch
// spectrum, its own pruning radius), the C = 32 transformed rows stay in LDS as real values, and thread (triangle, part)
// adds f_a f_b f_c over its part of the row's N cells before the workgroup moves to its next (x, y) position.  The real cubes
// never exist.  Products of fp32 values in fp32, running sums in double, per-thread partials reduced in a fixed order
// (deterministic).  Same transform as rows_c2r_kernel (Z[k] from the half spectrum, conj-FFT-conj in two register stages).
constexpr int TRI_ROWS = 32;                 // shells per workgroup = rows of the batched C2R
struct TriShells { const float2* work[TRI_ROWS]; int kmax[TRI_ROWS]; int count; };

template <int R1, int R2>
__global__ void __launch_bounds__(TRI_ROWS * R2)
#ifndef TRI_OCC_FREE
__attribute__((amdgpu_waves_per_eu(4, 4)))          // 128 VGPRs: two 512-thread workgroups per CU (74 KB of LDS each) up to n = 512
#endif
rows_c2r_triangles_kernel(TriShells sh, const float2* __restrict__ tw_g, size_t nrows, size_t in_pitch, const int* __restrict__ tri,
                          int ntri, int parts, double* __restrict__ partial) {
    constexpr int C = TRI_ROWS, M = R1 * R2, N = 2 * M, NT = C * R2, R2P = R2 + 1, MP = M + 1;
    extern __shared__ float2 lds[];
    float2* Y = lds;                     // X rows [r][k <= M], then the stage buffer, then the real rows [r][2 j, 2 j + 1]
    float2* tw = lds + C * (R1 * R2P > MP ? R1 * R2P : MP);
    __shared__ const float2* wsrc[C];
    __shared__ int wkmax[C];
    for (int i = threadIdx.x; i < N; i += NT) tw[i] = tw_g[i];
    if (threadIdx.x < C) {
        const int r = (int)threadIdx.x;
        wsrc[r] = sh.work[r < sh.count ? r : sh.count - 1];
        wkmax[r] = r < sh.count ? sh.kmax[r] : 0;             // rows past the last shell: zeros
    }
    // triangle (t, part) of this thread; fields as FLOAT offsets into Y (row pitch 2 MP floats)
    const int t = threadIdx.x % ntri, part = threadIdx.x / ntri;
    const bool worker = part < parts;
    int ia = 0, ib = 0, ic = 0;
    if (worker) {
        ia = tri[3 * t] * (2 * MP); ib = tri[3 * t + 1] * (2 * MP); ic = tri[3 * t + 2] * (2 * MP);
        if (ib == ic) { const int a = ia; ia = ib; ic = a; }          // a repeated field first: two LDS reads per term
        else if (ia == ic) { const int b = ib; ib = ia; ic = b; }
    }
    const bool wave_pairs = __all((int)(!worker || ia == ib)) != 0;
    // (an even number of cells per part: the products below read the rows two cells - one float2 - at a time)
    const int per = (((N + parts - 1) / parts) + 1) & ~1, c_lo = min(part * per, N), c_hi = min(c_lo + per, N);
    double acc = 0.0;
    __syncthreads();
    const int rr = threadIdx.x / R2, q = threadIdx.x % R2;        // load task: row rr, k = q + R2 i
    const int kmax = wkmax[rr];
    const float2* const wrow = wsrc[rr];
    float2 tmp[R1 + 1];
    auto fetch = [&](size_t pos) {
        const float2* src = wrow + pos * in_pitch;
        // (the clamped indices do not depend on the position either: hoisted, their 17 64-bit offsets cost 51 VGPRs)
        int km = kmax > 0 ? kmax - 1 : 0;
        asm volatile("" : "+v"(km));
#pragma unroll
        for (int i = 0; i <= R1; ++i) {
            const unsigned k = (unsigned)min(q + R2 * i, km);     // unconditional load from inside the written part
            tmp[i] = src[k];
        }
    };
    constexpr bool PREFETCH = R2 == 16;        // (1024 threads at n = 1024 leave 128 VGPRs: no second row set in registers)
    if (PREFETCH && (size_t)blockIdx.x < nrows) fetch(blockIdx.x);
#pragma clang loop unroll(disable)
    for (size_t pos = blockIdx.x; pos < nrows; pos += gridDim.x) {
        if (!PREFETCH) fetch(pos);
        int kmv = kmax;
        asm volatile("" : "+v"(kmv));
#pragma unroll
        for (int i = 0; i <= R1; ++i) {
            const int k = q + R2 * i;
            if (k <= M) Y[rr * MP + k] = k < kmv ? tmp[i] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        // (the twiddles a thread uses do not depend on the position: left to itself the compiler keeps all 2 R1 of them in
        // registers across the whole loop - 64 VGPRs, 104 spilled at the 128 this kernel may use; re-read from LDS instead)
        int n2 = threadIdx.x % R2;
        asm volatile("" : "+v"(n2));
        const int r = threadIdx.x / R2;
        float2 v[R1];
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1) {
            const int k = n1 * R2 + n2;
            const float2 xk = Y[r * MP + k], xm = Y[r * MP + M - k];
            const float2 e = make_float2(0.5f * (xk.x + xm.x), 0.5f * (xk.y - xm.y));
            const float2 d = make_float2(0.5f * (xk.x - xm.x), 0.5f * (xk.y + xm.y));
            const float2 w = tw[k];
            const float2 tt = make_float2(d.x * w.x + d.y * w.y, d.y * w.x - d.x * w.y);
            v[n1] = make_float2(e.x - tt.y, -(e.y + tt.x));
        }
        __syncthreads();
        fft_reg<R1>(v);
#pragma unroll
        for (int k1 = 0; k1 < R1; ++k1) {
            float2 y = v[bitrev(k1, ilog2(R1))];
            if (k1 != 0) y = cmul(y, tw[2 * n2 * k1]);
            Y[(r * R1 + k1) * R2P + n2] = y;
        }
        __syncthreads();
        float2 u[R2];
        const int k1 = threadIdx.x % R1, r2 = threadIdx.x / R1;
        const bool task2 = r2 < C;
        if (task2) {
#pragma unroll
            for (int j = 0; j < R2; ++j) u[j] = Y[(r2 * R1 + k1) * R2P + j];
            fft_reg<R2>(u);
        }
        __syncthreads();
        if (task2) {
            // z[j] = conj(result[j]); x[2j] = 2 Re z, x[2j+1] = 2 Im z: the row's real values / 2, in place
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
                const float2 zz = u[bitrev(k2, ilog2(R2))];
                Y[r2 * MP + k1 + R1 * k2] = make_float2(zz.x, -zz.y);
            }
        }
        __syncthreads();
        // the next position's rows: in flight across the product loop (the longest phase, and the one with the fewest live
        // registers - held across the two register FFTs the 34 extra VGPRs spilled)
        if (PREFETCH && pos + gridDim.x < nrows) fetch(pos + gridDim.x);
#ifndef TRI_FUSE_ABLATE
#define TRI_FUSE_ABLATE 0           // perf experiments: 1 = no products, 2 = no global loads, 4 = no transform
#endif
        if (worker && !(TRI_FUSE_ABLATE & 1)) {
            // two cells per LDS read; the part's <= N / parts products are summed in fp32 (each product carries 6e-8 already;
            // ~90 terms add 6e-7 of the PART's sum, random from part to part), the parts and positions in double
            const float* f = reinterpret_cast<const float*>(Y);
            float s0 = 0.f, s1 = 0.f;
            if (wave_pairs) {
#pragma unroll 4
                for (int c = c_lo; c < c_hi; c += 2) {
                    const float2 x = *reinterpret_cast<const float2*>(f + ia + c), y = *reinterpret_cast<const float2*>(f + ic + c);
                    s0 = fmaf(x.x * x.x, y.x, s0);
                    s1 = fmaf(x.y * x.y, y.y, s1);
                }
            } else {
#pragma unroll 4
                for (int c = c_lo; c < c_hi; c += 2) {
                    const float2 x = *reinterpret_cast<const float2*>(f + ia + c), y = *reinterpret_cast<const float2*>(f + ib + c);
                    const float2 z = *reinterpret_cast<const float2*>(f + ic + c);
                    s0 = fmaf(x.x * y.x, z.x, s0);
                    s1 = fmaf(x.y * y.y, z.y, s1);
                }
            }
            acc += (double)s0 + (double)s1;
        }
        __syncthreads();                                          // the rows are overwritten by the next position
    }
    partial[(size_t)blockIdx.x * NT + threadIdx.x] = worker ? acc : 0.0;
}

// out[t] = factor * sum over workgroups and parts of partial[block][part * ntri + t], fixed order
__global__ void __launch_bounds__(256)
triangles_reduce_kernel(const double* __restrict__ partial, int nblocks, int nthreads, int ntri, int parts, double factor,
                        double* __restrict__ out) {
    const int t = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks * parts; i += 256) {
        const int blk = i / parts, part = i % parts;
        acc += partial[(size_t)blk * nthreads + part * ntri + t];
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __shared__ double w[4];
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[t] = factor * ((w[0] + w[1]) + (w[2] + w[3]));
}

// -------------------------------------------------- low-k side channel in double
// The fp32 transform leaves a white round-off floor of ~1e-7 of the field's rms amplitude on every mode.  A
// spectrum with a large dynamic range (the cold lattice of the bench: the lowest shells hold 1e-5 of the peak
// power) then misses the 1e-6 target on its lowest shells by 2e-6 / |m|^2.  The few modes with |m_i| <= MLOW are
// therefore ALSO evaluated as plain DFT sums in double - one more read of the grid, ~12 fp64 FMA per cell - and
// the shells they fill completely (|m| < MLOW + 1) take their sums from here:
//   z  one wave per (x, y) row: lane l holds z = l + 64 j; the j sum uses the (N/64)-th roots of unity as
//      compile-time constants, one twiddle e^{-2 pi i kz l / N} per lane and kz, then a fixed xor-shuffle tree;
//   y  per x plane, sums over y for ky = -MLOW..MLOW;   x  sums over x for kx = -MLOW..MLOW, then the shells.
// Deterministic (fixed summation trees).  The grid may hold rho or rho - mean: only the unused DC mode differs.
constexpr int MLOW = 5;                  // shells 0..4 (|m| in [1, 6), by the binning rule) are taken from the double-precision sums
constexpr int MBOX = MLOW + 1;           // modes |m_i| <= MBOX are evaluated: under the float64 rule a vector of norm exactly 6 may fall into shell 4
// e^{-2 pi i r / 32} for odd r: (kC32o[r / 2], kS32o[r / 2]) - cos and -sin of pi r / 16
__device__ constexpr double kC32o[16] = {0.98078528040323044, 0.83146961230254524, 0.55557023301960222, 0.19509032201612827,
                                         -0.19509032201612827, -0.55557023301960222, -0.83146961230254524, -0.98078528040323044,
                                         -0.98078528040323044, -0.83146961230254524, -0.55557023301960222, -0.19509032201612827,
                                         0.19509032201612827, 0.55557023301960222, 0.83146961230254524, 0.98078528040323044};
__device__ constexpr double kS32o[16] = {-0.19509032201612827, -0.55557023301960222, -0.83146961230254524, -0.98078528040323044,
                                         -0.98078528040323044, -0.83146961230254524, -0.55557023301960222, -0.19509032201612827,
                                         0.19509032201612827, 0.55557023301960222, 0.83146961230254524, 0.98078528040323044,
                                         0.98078528040323044, 0.83146961230254524, 0.55557023301960222, 0.19509032201612827};
// e^{-2 pi i r / 16} = (kC16[r], kS16[r])
__device__ constexpr double kC16[16] = {1.0, 0.92387953251128674, 0.70710678118654752, 0.38268343236508977, 0.0,
                                        -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                        -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                        0.38268343236508977, 0.70710678118654752, 0.92387953251128674};
__device__ constexpr double kS16[16] = {0.0, -0.38268343236508977, -0.70710678118654752, -0.92387953251128674, -1.0,
                                        -0.92387953251128674, -0.70710678118654752, -0.38268343236508977, 0.0,
                                        0.38268343236508977, 0.70710678118654752, 0.92387953251128674, 1.0,
                                        0.92387953251128674, 0.70710678118654752, 0.38268343236508977};

// One level of a TRANSPOSED wave reduction: the lanes whose `mask` bit is clear keep the lower half of the `count`
// running sums, the others the upper half, and each adds its partner's half - half the work of a plain xor tree at
// every level (56 element steps for 28 sums instead of 168).
template <int COUNT>
__device__ inline void reduce_level(double (&v)[28], int lane, int mask) {
    const bool upper = (lane & mask) != 0;
#pragma unroll
    for (int i = 0; i < COUNT / 2; ++i) {
        const double keep = upper ? v[i + COUNT / 2] : v[i];
        const double send = upper ? v[i] : v[i + COUNT / 2];
        v[i] = keep + __shfl_xor(send, mask, 64);
    }
}

template <int NJ, int FOLDW>
__global__ void __launch_bounds__(256)
lowk_z_kernel(const float* __restrict__ grid, const float* __restrict__ rec, int n, size_t nrows, double2* __restrict__ out) {
    static_assert(MBOX == 6, "the reduction below is laid out for 2 rows x 7 kz x (re, im) = 28 sums");
    const int lane = threadIdx.x & 63;
    // the lane's twiddles e^{-2 pi i kz lane / n}: once per wave, which then walks many row pairs
    double twc[MBOX + 1], tws[MBOX + 1];
#pragma unroll
    for (int kz = 0; kz <= MBOX; ++kz) sincospi(-2.0 * (double)((kz * lane) % n) / (double)n, &tws[kz], &twc[kz]);
    const size_t npairs = nrows / 2, nwaves = (size_t)gridDim.x * 4;          // nrows is even
    for (size_t pair = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); pair < npairs; pair += nwaves) {
        double v[28];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const size_t row = 2 * pair + rr;
            const float* in = grid + row * (size_t)n;
            float fr[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fr[j] = in[lane + 64 * j];
            if (FOLDW != 0) {
                constexpr int W = FOLDW != 0 ? FOLDW : 2;
                const float* src[3];
                const int ns = ast::halo_sources<float, W>(rec, (int)(row / n), (int)(row % n), n, n / ast::TX, n / ast::TY, src);
                if (ns > 0) {                            // the records first, then onto the row: the fold's order,
                    float h[NJ];                         // i.e. the fp32 value the FFT's z pass sees
#pragma unroll
                    for (int j = 0; j < NJ; ++j) h[j] = src[0][lane + 64 * j];
                    if (ns > 1) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) h[j] += src[1][lane + 64 * j];
                    }
                    if (ns > 2) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j) h[j] += src[2][lane + 64 * j];
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) fr[j] += h[j];
                }
            }
            double f[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) f[j] = (double)fr[j];
#pragma unroll
            for (int kz = 0; kz <= MBOX; ++kz) {
                double re = 0.0, im = 0.0;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    if (NJ <= 16) {
                        constexpr int step = NJ <= 16 ? 16 / NJ : 1;
                        const int r = ((kz * j) % NJ) * step;    // e^{-2 pi i (kz j) / NJ}: a compile-time constant after unrolling
                        if (r == 0) re += f[j];
                        else if (r == 4) im -= f[j];
                        else if (r == 8) re -= f[j];
                        else if (r == 12) im += f[j];
                        else { re += f[j] * kC16[r]; im += f[j] * kS16[r]; }
                    } else {                                     // NJ = 32 (n = 2048): 32nd roots of unity
                        const int r = (kz * j) % 32;
                        if (r == 0) re += f[j];
                        else if (r == 8) im -= f[j];
                        else if (r == 16) re -= f[j];
                        else if (r == 24) im += f[j];
                        else if (r % 2 == 0) { re += f[j] * kC16[r / 2]; im += f[j] * kS16[r / 2]; }
                        else { re += f[j] * kC32o[r / 2]; im += f[j] * kS32o[r / 2]; }
                    }
                }
                v[rr * 14 + 2 * kz] = re * twc[kz] - im * tws[kz];
                v[rr * 14 + 2 * kz + 1] = re * tws[kz] + im * twc[kz];
            }
        }
        // 28 sums over the 64 lanes: 28 -> 14 -> 7 (+1 zero) -> 4 -> 2 -> 1, then the last pair
        reduce_level<28>(v, lane, 32);
        reduce_level<14>(v, lane, 16);
        v[7] = 0.0;
        reduce_level<8>(v, lane, 8);
        reduce_level<4>(v, lane, 4);
        reduce_level<2>(v, lane, 2);
        v[0] += __shfl_xor(v[0], 1, 64);
        // lane bits 5..1 say which sum it holds: element 14 b5 + 7 b4 + (4 b3 + 2 b2 + b1), the last term 7 being the pad
        const int sub = ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        if ((lane & 1) == 0 && sub < 7) {
            const int e = ((lane >> 5) & 1) * 14 + ((lane >> 4) & 1) * 7 + sub;      // = row-in-pair * 14 + 2 kz + (re | im)
            reinterpret_cast<double*>(out)[(2 * pair + e / 14) * (size_t)(2 * (MBOX + 1)) + e % 14] = v[0];
        }
    }
}

// One axis of the low-k transform: in[(outer * nt + t) * inner + i], t < nt, is summed with e^{-2 pi i k (t0 + t) / n} for
// k = -MBOX..MBOX.  Block (outer, part) adds up the t of part `part` (of gridDim.y), four independent accumulators so
// that several loads are in flight, and writes out[((outer * nparts + part) * (2 MBOX + 1) + k + MBOX) * inner + i].
// y pass: outer = x plane, inner = MBOX + 1, one part.  x pass: outer = 1, inner = (2 MBOX + 1)(MBOX + 1), LOWK_XPARTS
// parts, then lowk_parts_reduce_kernel adds the parts in order.
constexpr int LOWK_XPARTS = 64;
__global__ void __launch_bounds__(256)
lowk_axis_kernel(const double2* __restrict__ in, int n, int nt, int t0, int inner, double2* __restrict__ out) {
    extern __shared__ double2 tw[];                       // e^{-2 pi i t / n}
    for (int t = threadIdx.x; t < n; t += 256) {
        double sn, cs;
        sincospi(-2.0 * (double)t / (double)n, &sn, &cs);
        tw[t] = make_double2(cs, sn);
    }
    __syncthreads();
    const size_t outer = blockIdx.x;
    const int nparts = gridDim.y, part = blockIdx.y;
    const int per = (nt + nparts - 1) / nparts, ta = part * per, tb = min(nt, ta + per);
    const int nout = (2 * MBOX + 1) * inner;
    if (nout <= 128) {
        // few outputs (the y pass: 91): 256 / nout thread groups share the t range, interleaved, and their partial
        // sums are added in group order
        __shared__ double2 part_sum[256];
        const int groups = 256 / nout, o = threadIdx.x % nout, gq = threadIdx.x / nout;
        double re = 0.0, im = 0.0, re2 = 0.0, im2 = 0.0;
        if (gq < groups) {
            const int k = o / inner - MBOX, i = o % inner;
            int t = ta + gq;
            for (; t + groups < tb; t += 2 * groups) {           // two loads in flight
                const double2 v = in[(outer * nt + t) * inner + i], v2 = in[(outer * nt + t + groups) * inner + i];
                const double2 w = tw[(unsigned)(k * (t0 + t)) & (unsigned)(n - 1)];
                const double2 w2 = tw[(unsigned)(k * (t0 + t + groups)) & (unsigned)(n - 1)];
                re += v.x * w.x - v.y * w.y;
                im += v.x * w.y + v.y * w.x;
                re2 += v2.x * w2.x - v2.y * w2.y;
                im2 += v2.x * w2.y + v2.y * w2.x;
            }
            if (t < tb) {
                const double2 v = in[(outer * nt + t) * inner + i];
                const double2 w = tw[(unsigned)(k * (t0 + t)) & (unsigned)(n - 1)];
                re += v.x * w.x - v.y * w.y;
                im += v.x * w.y + v.y * w.x;
            }
        }
        part_sum[threadIdx.x] = make_double2(re + re2, im + im2);
        __syncthreads();
        if (gq == 0) {
            double sr = 0.0, si = 0.0;
            for (int q = 0; q < groups; ++q) { sr += part_sum[q * nout + o].x; si += part_sum[q * nout + o].y; }
            const int k = o / inner - MBOX, i = o % inner;
            out[((outer * nparts + part) * (2 * MBOX + 1) + k + MBOX) * inner + i] = make_double2(sr, si);
        }
        return;
    }
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int k = o / inner - MBOX, i = o % inner;
        double re[4] = {0.0, 0.0, 0.0, 0.0}, im[4] = {0.0, 0.0, 0.0, 0.0};
        for (int tq = ta; tq < tb; tq += 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = min(tq + q, tb - 1);        // unconditional load; the duplicate is not added
                const double2 v = in[(outer * nt + t) * inner + i];
                const double2 w = tw[(unsigned)(k * (t0 + t)) & (unsigned)(n - 1)];       // n is a power of two
                if (tq + q < tb) {
                    re[q] += v.x * w.x - v.y * w.y;
                    im[q] += v.x * w.y + v.y * w.x;
                }
            }
        }
        out[((outer * nparts + part) * (2 * MBOX + 1) + k + MBOX) * inner + i] =
            make_double2((re[0] + re[1]) + (re[2] + re[3]), (im[0] + im[1]) + (im[2] + im[3]));
    }
}

// acc[i] (+)= sum over parts of in[part * count + i], parts in index order
__global__ void __launch_bounds__(256)
lowk_parts_reduce_kernel(const double2* __restrict__ in, int nparts, int count, int accumulate, double2* __restrict__ acc) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    double re = accumulate ? acc[i].x : 0.0, im = accumulate ? acc[i].y : 0.0;
    for (int p = 0; p < nparts; ++p) {
        re += in[(size_t)p * count + i].x;
        im += in[(size_t)p * count + i].y;
    }
    acc[i] = make_double2(re, im);
}

// modes[kx + MBOX][ky + MBOX][kz] -> sums[s] = pnorm * sum w |delta_k|^2 over the modes of shell s < MLOW (shell by
// the binning rule in force): one thread per mode works out (shell, weighted power), then one thread per shell adds
// its modes in index order
__global__ void __launch_bounds__(256)
lowk_shell_kernel(const double2* __restrict__ modes, double pnorm, double kf_rule, double* __restrict__ sums) {
    constexpr int NM = (2 * MBOX + 1) * (2 * MBOX + 1) * (MBOX + 1);
    __shared__ double val[NM];
    __shared__ signed char sh[NM];
    for (int i = threadIdx.x; i < NM; i += 256) {
        const int c = i % (MBOX + 1), b = (i / (MBOX + 1)) % (2 * MBOX + 1) - MBOX, a = i / ((MBOX + 1) * (2 * MBOX + 1)) - MBOX;
        const int m2 = a * a + b * b + c * c;
        int r = (int)sqrt((double)m2);
        while (r * r > m2) --r;
        while ((r + 1) * (r + 1) <= m2) ++r;
        if (kf_rule != 0.0 && r * r == m2 && r > 0) r = ast::float64_edge_norm(r, a, b, c, kf_rule);
        const double2 v = modes[i];
        sh[i] = (signed char)(r - 1 < MLOW ? r - 1 : -1);
        val[i] = (c > 0 ? 2.0 : 1.0) * (v.x * v.x + v.y * v.y);
    }
    __syncthreads();
    if ((int)threadIdx.x < MLOW) {
        double acc = 0.0;
        for (int i = 0; i < NM; ++i)
            if (sh[i] == (int)threadIdx.x) acc += val[i];
        sums[threadIdx.x] = acc * pnorm;
    }
}

__global__ void lowk_patch_kernel(const double* __restrict__ sums, int count, double* __restrict__ psum) {
    if ((int)threadIdx.x < count) psum[threadIdx.x] += sums[threadIdx.x];
}

// ------------------------------------------------------------------ host side
// The fall words of the fused x pass (see strided_c2c_kernel): for workgroup (ky = b, kz tile) and thread (c, sub) bit k2
// covers the mode (kx row sub + R1 k2, ky, kz = 16 tile + c).  Same thread geometry as launch_c2c<.., true>.
template <int R1, int R2, int C>
__global__ void edge_fall_kernel(unsigned* __restrict__ out, unsigned tiles_per_batch, int ncols, double kf) {
    constexpr int N = R1 * R2;
    const unsigned tile = blockIdx.x % tiles_per_batch, b = blockIdx.x / tiles_per_batch;
    const int c = threadIdx.x % C, sub = threadIdx.x / C;
    const int kz = (int)(tile * C + c), ky = (int)b > N / 2 ? (int)b - N : (int)b;
    unsigned word = 0;
    if (sub < R1 && kz < ncols) {
        for (int k2 = 0; k2 < R2; ++k2) {
            const int row = sub + R1 * k2, kx = row > N / 2 ? row - N : row;
            const int m2 = kx * kx + ky * ky + kz * kz;
            int r = (int)sqrt((double)m2);
            while (r * r > m2) --r;
            while ((r + 1) * (r + 1) <= m2) ++r;
            if (r * r == m2 && r > 0 && ast::float64_edge_norm(r, kx, ky, kz, kf) != r) word |= 1u << k2;
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = word;
}

struct EdgeFallCache {
    std::mutex m;
    struct Entry { int dev; size_t n; double boxsize; unsigned* words; };
    std::vector<Entry> tabs;
    const unsigned* get(size_t n, double boxsize, hipStream_t s) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& t : tabs) if (t.dev == dev && t.n == n && t.boxsize == boxsize) return t.words;
        const size_t nz = n / 2 + 1, tiles = (nz + 15) / 16;
        const unsigned nt = n == 256 ? 256 : 512;                 // threads of launch_c2c<R1, R2, 16, true>
        unsigned* d = nullptr;
        if (hipMalloc(&d, n * tiles * nt * sizeof(unsigned)) != hipSuccess) return nullptr;
        const double kf = 2.0 * M_PI / boxsize;
        const unsigned blocks = (unsigned)(n * tiles);
        if (n == 1024) edge_fall_kernel<32, 32, 16><<<blocks, nt, 0, s>>>(d, (unsigned)tiles, (int)nz, kf);
        else if (n == 512) edge_fall_kernel<16, 32, 16><<<blocks, nt, 0, s>>>(d, (unsigned)tiles, (int)nz, kf);
        else edge_fall_kernel<16, 16, 16><<<blocks, nt, 0, s>>>(d, (unsigned)tiles, (int)nz, kf);
        if (hipGetLastError() != hipSuccess) return nullptr;
        if (tabs.size() >= 4) { (void)hipFree(tabs.front().words); tabs.erase(tabs.begin()); }      // a few (N, L) at most
        tabs.push_back({dev, n, boxsize, d});
        return d;
    }
} g_edge;

struct TwiddleCache {
    std::mutex m;
    std::vector<std::pair<std::pair<int, int>, float2*>> tabs;   // (device, N) -> table
    float2* get(int n) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& t : tabs) if (t.first.first == dev && t.first.second == n) return t.second;
        std::vector<float2> h(n);
        for (int i = 0; i < n; ++i) {
            const double a = -2.0 * M_PI * (double)i / (double)n;
            h[i] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        float2* d = nullptr;
        if (hipMalloc(&d, n * sizeof(float2)) != hipSuccess) return nullptr;
        if (hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        tabs.push_back({{dev, n}, d});
        return d;
    }
} g_tw;

template <int R1, int R2, int C, bool POWER>
int launch_c2c(float2* data, const float2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride,
               float scale, double* partial, hipStream_t s, const unsigned* edge_fall = nullptr, int ky0 = 0, long long prune2 = 0,
               PackDst disc = PackDst{}) {
    constexpr int N = R1 * R2, NT = C * (R1 > R2 ? R1 : R2);
    constexpr bool SPLIT = (POWER || C > 16) && R1 == R2 && N * C * sizeof(float2) > 64 * 1024;       // as in the kernel
    const size_t lds = (size_t)((SPLIT ? N / 2 : N) * C + N) * sizeof(float2) + (POWER ? (N / 2) * sizeof(double) : 0);
    static ast::PerDeviceOnce attr_once;
    if (attr_once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&strided_c2c_kernel<R1, R2, C, POWER>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark();
    }
    const size_t tiles = (ncols + C - 1) / C;
    AST_CHECK_ARG(tiles * batch < 0x7fffffffull);
    AST_CHECK_ARG((size_t)(R2 - 1) * elem_stride + ncols < (1ull << 29));      // the kernel's 32-bit lane offsets
    strided_c2c_kernel<R1, R2, C, POWER><<<(unsigned)(tiles * batch), NT, lds, s>>>(data, tw, elem_stride, ncols,
                                                                                   batch_stride, (unsigned)tiles, scale,
                                                                                   partial, edge_fall, ShellMask{nullptr, 0, prune2, ky0}, disc);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

template <int R1, int R2, int C>
int launch_c2c_inv(float2* data, const float2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride,
                   float scale, ShellMask mask, hipStream_t s, ShellBatch sb = ShellBatch{}) {
    constexpr int N = R1 * R2, NT = C * (R1 > R2 ? R1 : R2);
    const size_t lds = (size_t)(N * C + N) * sizeof(float2);
    static ast::PerDeviceOnce attr_once;
    if (attr_once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&strided_c2c_kernel<R1, R2, C, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark();
    }
    const size_t tiles = (ncols + C - 1) / C;
    AST_CHECK_ARG(tiles * batch < 0x7fffffffull);
    AST_CHECK_ARG((size_t)(R2 - 1) * elem_stride + ncols < (1ull << 29));      // the kernel's 32-bit lane offsets
    strided_c2c_kernel<R1, R2, C, false, true><<<dim3((unsigned)(tiles * batch), (unsigned)(sb.count > 0 ? sb.count : 1)), NT, lds, s>>>(
        data, tw, elem_stride, ncols, batch_stride, (unsigned)tiles, scale, nullptr, nullptr, mask, PackDst{}, sb);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

template <int R1, int R2, int C, int PACKM = 1>
int launch_c2c_pack(float2* data, const float2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride,
                    float scale, PackDst pack, hipStream_t s) {
    constexpr int N = R1 * R2, NT = C * (R1 > R2 ? R1 : R2);
    constexpr bool SPLIT = C > 16 && R1 == R2 && N * C * sizeof(float2) > 64 * 1024;       // as in the kernel
    const size_t lds = (size_t)((SPLIT ? N / 2 : N) * C + N) * sizeof(float2);
    static ast::PerDeviceOnce attr_once;
    if (attr_once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&strided_c2c_kernel<R1, R2, C, false, false, PACKM>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark();
    }
    const size_t tiles = (ncols + C - 1) / C;
    AST_CHECK_ARG(tiles * batch < 0x7fffffffull);
    AST_CHECK_ARG((size_t)(R2 - 1) * elem_stride + ncols < (1ull << 29));      // the kernel's 32-bit lane offsets
    strided_c2c_kernel<R1, R2, C, false, false, PACKM><<<(unsigned)(tiles * batch), NT, lds, s>>>(
        data, tw, elem_stride, ncols, batch_stride, (unsigned)tiles, scale, nullptr, nullptr, ShellMask{nullptr, 0, 0}, pack);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

int dispatch_c2c_inv(size_t n, float2* d, const float2* tw, size_t elem_stride, size_t ncols, size_t batch, size_t batch_stride,
                     float scale, ShellMask mask, hipStream_t s, ShellBatch sb = ShellBatch{}) {
    if (n == 1024) return launch_c2c_inv<32, 32, 16>(d, tw, elem_stride, ncols, batch, batch_stride, scale, mask, s, sb);
    if (n == 512) return launch_c2c_inv<16, 32, 16>(d, tw, elem_stride, ncols, batch, batch_stride, scale, mask, s, sb);
    return launch_c2c_inv<16, 16, 16>(d, tw, elem_stride, ncols, batch, batch_stride, scale, mask, s, sb);
}

template <bool POWER>
int dispatch_c2c(size_t n, float2* d, const float2* tw, size_t elem_stride, size_t ncols, size_t batch,
                 size_t batch_stride, float scale, double* partial, hipStream_t s, const unsigned* edge_fall = nullptr, int ky0 = 0,
                 long long prune2 = 0, PackDst disc = PackDst{}) {
#ifndef FWD_C
#define FWD_C 32                    // columns per workgroup of the plain forward pass at N = 1024: 32 = 256-byte row pieces, one 1024-thread
                                    // workgroup per CU with the split exchange (y pass 2.02 -> 1.91 ms); 16 = as in rounds 1-2
#endif
    if constexpr (!POWER) { if (n == 1024) return launch_c2c<32, 32, FWD_C, false>(d, tw, elem_stride, ncols, batch, batch_stride, scale, partial, s, edge_fall, ky0, prune2); }
    if (n == 1024) return launch_c2c<32, 32, 16, POWER>(d, tw, elem_stride, ncols, batch, batch_stride, scale, partial, s, edge_fall, ky0, prune2, disc);
    if (n == 512) return launch_c2c<16, 32, 16, POWER>(d, tw, elem_stride, ncols, batch, batch_stride, scale, partial, s, edge_fall, ky0, prune2, disc);
    return launch_c2c<16, 16, 16, POWER>(d, tw, elem_stride, ncols, batch, batch_stride, scale, partial, s, edge_fall, ky0, prune2, disc);
}

template <int R1, int R2, int C, int FOLDW = 0, bool LOWK = false>
int launch_r2c(const float* in, float2* out, const float2* tw, size_t nrows, size_t in_pitch, size_t out_pitch,
               float scale, float mean, hipStream_t s, const float* rec = nullptr, const double2* lowk_lane = nullptr,
               double* lowz = nullptr, SlabFold sf = SlabFold{}) {
    constexpr int M = R1 * R2, N = 2 * M, NT = C * R2;
    constexpr int BUF = C * (R1 * (R2 + 1) > M + 1 ? R1 * (R2 + 1) : M + 1);
    const size_t lds = (size_t)(BUF + N) * sizeof(float2);
    static ast::PerDeviceOnce attr_once;
    if (attr_once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_r2c_kernel<R1, R2, C, FOLDW, LOWK>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark();
    }
    const size_t blocks = (nrows + C - 1) / C;
    AST_CHECK_ARG(blocks < 0x7fffffffull);
    rows_r2c_kernel<R1, R2, C, FOLDW, LOWK><<<(unsigned)blocks, NT, lds, s>>>(in, out, tw, nrows, in_pitch, out_pitch, scale, mean, rec,
                                                                             lowk_lane, lowz, sf);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

template <int R1, int R2, int C>
int launch_c2r(const float2* in, float* out, const float2* tw, size_t nrows, size_t in_pitch, size_t out_pitch, float scale,
               int kmax, hipStream_t s, C2RBatch cb = C2RBatch{}) {
    constexpr int M = R1 * R2, N = 2 * M, NT = C * R2;
    constexpr int BUF = C * (R1 * (R2 + 1) > M + 1 ? R1 * (R2 + 1) : M + 1);
    const size_t lds = (size_t)(BUF + N) * sizeof(float2);
    static ast::PerDeviceOnce attr_once;
    if (attr_once.need()) {
        AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_c2r_kernel<R1, R2, C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_once.mark();
    }
    const size_t blocks = (nrows + C - 1) / C;
    AST_CHECK_ARG(blocks < 0x7fffffffull);
    rows_c2r_kernel<R1, R2, C><<<dim3((unsigned)blocks, (unsigned)(cb.count > 0 ? cb.count : 1)), NT, lds, s>>>(in, out, tw, nrows, in_pitch, out_pitch,
                                                                                                        scale, kmax, cb);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ------------------------------------------------------------------ disc layout (host side)
// Built from (n, parts) alone - every rank computes the same table.  astrild_amd/slab.py restates it in Python (disc_layout)
// for the buffer sizes of the CPU doubles; tests compare the two.
constexpr int tile_r1(size_t n) { return n == 1024 ? 32 : 16; }           // rows per block = the first radix of the passes
struct DiscLayout {
    size_t n = 0;
    int parts = 0, r1 = 0;
    unsigned nblk = 0, tiles = 0;
    std::vector<unsigned> S, cumS;                 // plane size of each part (complex elements), sum of the lower parts'
    std::vector<unsigned short> gk;                // [part][j]: the part's row blocks in ascending order
    std::vector<unsigned char> part_of;            // [row block]
    std::vector<DiscEntry> tab;                    // [tile][row block]
    struct Dev { int dev; DiscEntry* tab; unsigned short* gk; };
    std::vector<Dev> devs;
    unsigned total() const { return cumS.back() + S.back(); }
};

static bool disc_row_in(size_t n, int row, unsigned tile) {
    const long long ky = row > (int)(n / 2) ? row - (long long)n : row, kz0 = 16ll * tile, h = (long long)(n / 2);
    return ky * ky + kz0 * kz0 <= h * h;           // the edge itself stays (float64 shell rule)
}

static void disc_build(DiscLayout& L, size_t n, int parts) {
    L.n = n; L.parts = parts; L.r1 = tile_r1(n);
    L.nblk = (unsigned)(n / L.r1); L.tiles = (unsigned)((n / 2 + 1 + 15) / 16);
    const unsigned nb = L.nblk / parts;
    std::vector<unsigned> area(L.nblk, 0);
    for (unsigned k = 0; k < L.nblk; ++k)
        for (int sub = 0; sub < L.r1; ++sub)
            for (unsigned t = 0; t < L.tiles; ++t) area[k] += disc_row_in(n, (int)(k * L.r1 + sub), t) ? 1u : 0u;
    // blocks in order of decreasing area (ties: ascending index), each to the part with the smallest sum so far among those
    // that still take blocks (ties: the lowest part): within 4 % of the mean at 8 parts, where contiguous ranges of rows
    // would leave the parts around k_y = 0 with 97 % of a full plane - and their links with as many bytes as before
    std::vector<unsigned> order(L.nblk);
    for (unsigned k = 0; k < L.nblk; ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](unsigned a, unsigned b) { return area[a] > area[b]; });
    std::vector<unsigned long long> sum(parts, 0);
    std::vector<unsigned> cnt(parts, 0);
    L.part_of.assign(L.nblk, 0);
    for (unsigned k : order) {
        int best = -1;
        for (int r = 0; r < parts; ++r)
            if (cnt[r] < nb && (best < 0 || sum[r] < sum[best])) best = r;
        L.part_of[k] = (unsigned char)best;
        sum[best] += area[k];
        ++cnt[best];
    }
    L.gk.clear();
    for (int r = 0; r < parts; ++r)
        for (unsigned k = 0; k < L.nblk; ++k)
            if (L.part_of[k] == r) L.gk.push_back((unsigned short)k);
    L.tab.assign((size_t)L.tiles * L.nblk, DiscEntry{0, 0, 0, 0});
    L.S.assign(parts, 0);
    L.cumS.assign(parts, 0);
    for (int r = 0; r < parts; ++r) {
        unsigned rows = 0;                          // valid (tile, row) pairs of the part so far, tile-major
        for (unsigned t = 0; t < L.tiles; ++t)
            for (unsigned j = 0; j < nb; ++j) {
                const unsigned k = L.gk[(size_t)r * nb + j];
                int lo = -1, hi = -1;
                for (int sub = 0; sub < L.r1; ++sub)
                    if (disc_row_in(n, (int)(k * L.r1 + sub), t)) { if (lo < 0) lo = sub; hi = sub + 1; }
                DiscEntry& e = L.tab[(size_t)t * L.nblk + k];
                if (lo < 0) { e.lohi = (unsigned)r << 16; continue; }
                e.offb = 16 * ((int)rows - lo);
                e.lohi = (unsigned)lo | (unsigned)hi << 8 | (unsigned)r << 16;
                rows += (unsigned)(hi - lo);        // (the rows of a block inside the disc are one range: n / 2 is a block boundary)
            }
        L.S[r] = 16 * rows;
    }
    for (int r = 1; r < parts; ++r) L.cumS[r] = L.cumS[r - 1] + L.S[r - 1];
    for (auto& e : L.tab) { const unsigned r = e.lohi >> 16; e.S = L.S[r]; e.cumS = L.cumS[r]; }
}

struct DiscCache {
    std::mutex m;
    std::vector<DiscLayout*> all;
    // host table (and, with `device`, its copy on the current device)
    const DiscLayout* get(size_t n, int parts, bool device, const DiscLayout::Dev** dv) {
        std::lock_guard<std::mutex> lock(m);
        DiscLayout* L = nullptr;
        for (auto* x : all) if (x->n == n && x->parts == parts) L = x;
        if (!L) { L = new DiscLayout; disc_build(*L, n, parts); all.push_back(L); }
        if (!device) return L;
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& d : L->devs) if (d.dev == dev) { *dv = &d; return L; }
        DiscLayout::Dev d{dev, nullptr, nullptr};
        if (hipMalloc(&d.tab, L->tab.size() * sizeof(DiscEntry)) != hipSuccess) return nullptr;
        if (hipMalloc(&d.gk, L->gk.size() * sizeof(unsigned short)) != hipSuccess) return nullptr;
        if (hipMemcpy(d.tab, L->tab.data(), L->tab.size() * sizeof(DiscEntry), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        if (hipMemcpy(d.gk, L->gk.data(), L->gk.size() * sizeof(unsigned short), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
        L->devs.reserve(64);
        L->devs.push_back(d);
        *dv = &L->devs.back();
        return L;
    }
} g_disc;

static bool disc_geometry_ok(size_t n, int parts) {
    return (n == 256 || n == 512 || n == 1024) && parts >= 1 && parts <= 256 && (n / tile_r1(n)) % (size_t)parts == 0;
}

}  // namespace

extern "C" int ast_fft_tile_supported(int dtype, size_t n) {
    return dtype == AST_F32 && (n == 256 || n == 512 || n == 1024) ? 1 : 0;
}

extern "C" int ast_fft_tile_c2c(void* data, int dtype, size_t n, size_t elem_stride, size_t ncols, size_t batch,
                                size_t batch_stride, double scale, void* stream) {
    AST_CHECK_ARG(data != nullptr && ncols >= 1 && batch >= 1 && elem_stride >= ncols);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_c2c: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("fft_tile.c2c", s);
    return dispatch_c2c<false>(n, (float2*)data, tw, elem_stride, ncols, batch, batch_stride, (float)scale, nullptr, s);
}

// The y pass of a rank's local planes with the slab pack fused into its stores: planes_d (nplanes, n, ncols) complex
// (rows k_y, the z pass's output) is transformed along k_y and written to packed_d as `parts` blocks of
// (nplanes, n / parts, ncols) - what ast_fft_tile_c2c followed by ast_slab_pack produce, without the extra read and
// write of the spectrum.  planes_d is left untouched.  parts: a power of two dividing n.  With self_out_d, part
// `self_part` (the rank's own piece) is written there as (nplanes, n / parts, ncols) - straight into the receive block -
// and its slot in packed_d stays unwritten (packed_d may be NULL when parts == 1).
extern "C" int ast_fft_tile_c2c_packed(const void* planes, void* packed, int dtype, size_t n, size_t ncols, size_t pitch,
                                       size_t nplanes, int parts, int self_part, void* self_out, double scale, void* stream) {
    AST_CHECK_ARG(planes != nullptr && planes != packed && planes != self_out && ncols >= 1 && nplanes >= 1 && pitch >= ncols);
    AST_CHECK_ARG((self_out == nullptr) || (self_part >= 0 && self_part < parts));
    AST_CHECK_ARG(packed != nullptr || (self_out != nullptr && parts == 1));
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    AST_CHECK_ARG(parts >= 1 && (parts & (parts - 1)) == 0 && n % (size_t)parts == 0);
    AST_CHECK_ARG(n / (size_t)parts >= (n == 1024 ? 32u : 16u));       // rows per part >= the first radix (uniform store bases)
    AST_CHECK_ARG(pitch < (1u << 24));
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_c2c_packed: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    PackDst pack;
    pack.out = (float2*)packed;
    pack.nbatch = (unsigned)nplanes;
    pack.pitch = (unsigned)pitch;
    if (self_out) { pack.self_out = (float2*)self_out; pack.self_part = (unsigned)self_part; }
    for (size_t c1 = n / (size_t)parts; c1 > 1; c1 >>= 1) ++pack.c1_log2;
    AST_PROF("fft_tile.c2c", s);
    float2* d = (float2*)const_cast<void*>(planes);
    // N = 1024: 16-column tiles.  Measured at the 8-rank slab shape (128 planes, rows pitched to whole lines, 528):
    // 16 columns 0.27 ms, 32 columns (256-byte row pieces, the split exchange of the single-GPU y pass; 8 VGPRs
    // spilled by the extra store addressing) 0.30 - AST_FFT_PACK_C32=1 selects it for A/B runs.  Unpitched rows (513):
    // 0.44-0.60 ms, every 128-byte piece straddling two lines.
    if (n == 1024 && pitch % 16 == 0 && getenv("AST_FFT_PACK_C32"))
        return launch_c2c_pack<32, 32, FWD_C>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
    if (n == 1024) return launch_c2c_pack<32, 32, 16>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
    if (n == 512) return launch_c2c_pack<16, 32, 16>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
    return launch_c2c_pack<16, 16, 16>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
}

// lane factors of the fused low-k z sums (rows_r2c_kernel<.., LOWK>): [n2 < R2][kz <= 6][A, B],
// A = e^{-2 pi i kz 2 n2 / N}, B = e^{-2 pi i kz (2 n2 + 1) / N}
__global__ void lowk_lane_kernel(double2* __restrict__ out, int n, int lanes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes * 7 * 2) return;
    const int p = i & 1, kz = (i >> 1) % 7, n2 = (i >> 1) / 7;
    double sn, cs;
    sincospi(-2.0 * (double)((kz * (2 * n2 + p)) % n) / (double)n, &sn, &cs);
    out[i] = make_double2(cs, sn);
}
struct LowkLaneCache {
    std::mutex m;
    struct E { int dev, n; double2* d; };
    std::vector<E> tabs;
    const double2* get(int n, hipStream_t s) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& t : tabs) if (t.dev == dev && t.n == n) return t.d;
        const int lanes = n == 1024 ? 32 : 16;
        double2* d = nullptr;
        if (hipMalloc(&d, (size_t)lanes * 7 * 2 * sizeof(double2)) != hipSuccess) return nullptr;
        lowk_lane_kernel<<<2, 256, 0, s>>>(d, n, lanes);
        if (hipGetLastError() != hipSuccess) return nullptr;
        tabs.push_back({dev, n, d});
        return d;
    }
} g_lowk_lane;

static int rows_r2c_impl(const void* in, void* out, int dtype, size_t n, size_t nrows, size_t in_pitch,
                         size_t out_pitch, double scale, double mean, void* stream, const void* rec = nullptr,
                         int window = 0, double* lowz = nullptr, SlabFold sf = SlabFold{}) {
    AST_CHECK_ARG(in != nullptr && out != nullptr && in != out && nrows >= 1);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    AST_CHECK_ARG(in_pitch >= n && in_pitch % 2 == 0 && out_pitch >= n / 2 + 1);
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_rows_r2c: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("fft_tile.rows_r2c", s);
    const float* i = (const float*)in;
    float2* o = (float2*)out;
    if (lowz != nullptr) {                   // with the low-k channel's z sums on the side
        const double2* lane = g_lowk_lane.get((int)n, s);
        if (!lane) { ast::set_error("ast_fft_tile_rows_r2c: low-k lane table allocation failed"); return AST_ERR_HIP; }
        const float* h = (const float*)rec;
        const float sc = (float)scale, mn = (float)mean;
        AST_CHECK_ARG(rec == nullptr || ((nrows == n * n || sf.ranges) && in_pitch == n && (window == AST_WIN_CIC || window == AST_WIN_TSC)));
        auto go = [&](auto r1, auto r2) {
            constexpr int R1 = decltype(r1)::value, R2 = decltype(r2)::value;
            if (rec == nullptr) return launch_r2c<R1, R2, 16, 0, true>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, nullptr, lane, lowz);
            if (window == AST_WIN_CIC) return launch_r2c<R1, R2, 16, 2, true>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, lane, lowz, sf);
            return launch_r2c<R1, R2, 16, 3, true>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, lane, lowz, sf);
        };
        if (n == 1024) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 32>{});
        if (n == 512) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{});
        return go(std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
    }
    if (rec != nullptr) {                    // fold the paint's halo records while loading (whole periodic grid, or a slab buffer's planes)
        AST_CHECK_ARG((nrows == n * n || sf.ranges) && in_pitch == n && (window == AST_WIN_CIC || window == AST_WIN_TSC));
        const float* h = (const float*)rec;
        const float sc = (float)scale, mn = (float)mean;
        if (window == AST_WIN_CIC) {
            if (n == 1024) return launch_r2c<16, 32, 16, 2>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
            if (n == 512) return launch_r2c<16, 16, 16, 2>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
            return launch_r2c<8, 16, 16, 2>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
        }
        if (n == 1024) return launch_r2c<16, 32, 16, 3>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
        if (n == 512) return launch_r2c<16, 16, 16, 3>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
        return launch_r2c<8, 16, 16, 3>(i, o, tw, nrows, in_pitch, out_pitch, sc, mn, s, h, nullptr, nullptr, sf);
    }
    if (n == 1024) return launch_r2c<16, 32, 16>(i, o, tw, nrows, in_pitch, out_pitch, (float)scale, (float)mean, s);
    if (n == 512) return launch_r2c<16, 16, 16>(i, o, tw, nrows, in_pitch, out_pitch, (float)scale, (float)mean, s);
    return launch_r2c<8, 16, 16>(i, o, tw, nrows, in_pitch, out_pitch, (float)scale, (float)mean, s);
}

// out_d[v] = the shell lookup's floor(sqrt(v)), v < count: lets the tests check it against integers
extern "C" int ast_fft_tile_isqrt_table(int* out, int count, void* stream) {
    AST_CHECK_ARG(out != nullptr && count > 0);
    tile_isqrt_table_kernel<<<1024, 256, 0, ast::as_stream(stream)>>>(out, count);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

extern "C" int ast_fft_tile_rows_r2c(const void* in, void* out, int dtype, size_t n, size_t nrows, size_t in_pitch,
                                     size_t out_pitch, double scale, void* stream) {
    return rows_r2c_impl(in, out, dtype, n, nrows, in_pitch, out_pitch, scale, 0.0, stream);
}

// The z pass of `nrows` rows that also leaves the low-k channel's z sums of those rows at lowz (nrows x 7 complex128,
// [row][kz]: what the first kernel of ast_lowk_modes would compute from a second read of the planes).
extern "C" int ast_fft_tile_rows_r2c_lowz(const void* in, void* out, int dtype, size_t n, size_t nrows, size_t in_pitch,
                                          size_t out_pitch, double scale, void* lowz, void* stream) {
    AST_CHECK_ARG(lowz != nullptr);
    return rows_r2c_impl(in, out, dtype, n, nrows, in_pitch, out_pitch, scale, 0.0, stream, nullptr, 0, (double*)lowz);
}

// The z pass of a plane range of a SLAB buffer painted by ast_paint_tiled_stage, folding the paint's halo records as the rows
// are loaded (what AST_PAINT_STAGE_FOLD would have added, in the same order: bit-identical): in_d = buffer plane xb0 (nrows =
// planes * n rows of n reals), halo_rec_d from ast_paint_tiled_halo for the same buffer of nx_alloc planes; only planes whose
// tile row lies in [fold_row_lo, fold_row_hi) are folded here (the rows that hold ghost planes are folded by the paint, before
// their planes travel).  lowz_d (optional): the low-k z sums of these rows, as ast_fft_tile_rows_r2c_lowz.
extern "C" int ast_fft_tile_rows_r2c_slab_halo(const void* in, void* out, int dtype, size_t n, size_t nrows, size_t out_pitch,
                                               double scale, const void* halo_rec, int window, int xb0, int nx_alloc,
                                               int fold_row_lo, int fold_row_hi, void* lowz, void* stream) {
    AST_CHECK_ARG(halo_rec != nullptr && nx_alloc >= 1 && xb0 >= 0 && nrows % n == 0 && xb0 + (int)(nrows / n) <= nx_alloc);
    AST_CHECK_ARG(fold_row_lo >= 0 && fold_row_hi >= fold_row_lo);
    SlabFold sf;
    sf.xb0 = xb0;
    sf.ntx = (size_t)nx_alloc == n ? 0 : (nx_alloc + ast::TX - 1) / ast::TX;        // a buffer of all n planes is the periodic grid
    sf.row_lo = fold_row_lo;
    sf.row_hi = fold_row_hi;
    sf.ranges = true;
    return rows_r2c_impl(in, out, dtype, n, nrows, n, out_pitch, scale, 0.0, stream, halo_rec, window, (double*)lowz, sf);
}

// The three passes of an (n, n, n) real -> (n, n, n/2+1) half-spectrum transform,
// delta_k = scale * sum_x f e^{-ikx}.  `out` is also the work array (y, x passes in place).
extern "C" int ast_fft_tile_r2c_3d(const void* in, void* out, int dtype, size_t n, double scale, void* stream) {
    AST_CHECK_ARG(in != nullptr && out != nullptr);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    const size_t nz = n / 2 + 1;
    int rc = ast_fft_tile_rows_r2c(in, out, dtype, n, n * n, n, nz, 1.0, stream);                   // z
    if (rc != AST_OK) return rc;
    rc = ast_fft_tile_c2c(out, dtype, n, nz, nz, n, n * nz, 1.0, stream);                          // y, per x-plane
    if (rc != AST_OK) return rc;
    return ast_fft_tile_c2c(out, dtype, n, n * nz, n * nz, 1, 0, scale, stream);                    // x
}

// Row pitch (in complex elements) of the scratch spectrum used by ast_fft_tile_power_3d:
// n/2+1 rounded up to a multiple of 16 so every 128-byte tile row is line-aligned.
extern "C" size_t ast_lowk_work_bytes(size_t n, size_t nx);
static size_t power_disc_bytes(size_t n) {          // the k_y pass's output in the disc layout (one part): 78 % of a spectrum
    const DiscLayout* L = disc_geometry_ok(n, 1) ? g_disc.get(n, 1, false, nullptr) : nullptr;
    return L ? ((size_t)n * L->S[0] * sizeof(float2) + 255) / 256 * 256 : 0;
}
static size_t power_core_bytes(size_t n) {
    const size_t nzp = ((n / 2 + 1) + 15) / 16 * 16;
    const size_t tiles = (n / 2 + 1 + 15) / 16;
    return n * n * nzp * sizeof(float2) + power_disc_bytes(n) + n * tiles * (n / 2 - 1) * sizeof(double) + 64 * sizeof(double);     // + low-k sums
}
static size_t lowk_area_bytes(size_t n);
extern "C" size_t ast_fft_tile_power_scratch_bytes(size_t n) {
    // spectrum, shell partials, low-k sums; then the low-k channel's own modes and work area (it runs on a second
    // stream beside the FFT passes, so it cannot borrow the spectrum's space any more)
    return (power_core_bytes(n) + 255) / 256 * 256 + lowk_area_bytes(n);
}

// FFTPower's shell sums of an (n, n, n) real grid without ever writing the spectrum:
// z pass (R2C) and y pass into `scratch`, x pass fused with the shell binning.
// psum_d[shell] += L^3 * sum_modes w |delta_k|^2, delta_k = rfftn(grid)/n^3  (auto power only).
static int power_3d_impl(const void* grid, void* scratch, size_t scratch_bytes, int dtype, size_t n, double boxsize,
                         double mean, double* psum, const void* rec, int window, int lowk, int binning, void* stream);

extern "C" int ast_fft_tile_power_3d(const void* grid, void* scratch, size_t scratch_bytes, int dtype, size_t n,
                                     double boxsize, double mean, int lowk, int binning, double* psum, void* stream) {
    return power_3d_impl(grid, scratch, scratch_bytes, dtype, n, boxsize, mean, psum, nullptr, 0, lowk, binning, stream);
}

// The same for a grid painted with AST_PAINT_OVERWRITE | AST_PAINT_DEFER_FOLD: `halo_rec` (from
// ast_paint_tiled_halo) is folded into the border rows as the z pass loads them.
extern "C" int ast_fft_tile_power_3d_halo(const void* grid, const void* halo_rec, int window, void* scratch,
                                          size_t scratch_bytes, int dtype, size_t n, double boxsize, double mean,
                                          int lowk, int binning, double* psum, void* stream) {
    AST_CHECK_ARG(halo_rec != nullptr && (window == AST_WIN_CIC || window == AST_WIN_TSC));
    return power_3d_impl(grid, scratch, scratch_bytes, dtype, n, boxsize, mean, psum, halo_rec, window, lowk, binning, stream);
}

constexpr size_t LOWK_MODES = (size_t)(2 * MBOX + 1) * (2 * MBOX + 1) * (MBOX + 1);
static size_t lowk_area_bytes(size_t n) { return (LOWK_MODES * sizeof(double2) + 255) / 256 * 256 + ast_lowk_work_bytes(n, n); }

// The low-k channel reads the grid only: it runs on a side stream while the main stream does the z and y passes (its z
// kernel is half arithmetic, the FFT passes are memory bound; its four small kernels hide completely).  One side
// stream and event pair per (device, main stream); the events are only ever recorded in that main stream's order.
struct SideStream { int dev; hipStream_t main, side; hipEvent_t in, done; };
struct SideStreamCache {
    std::mutex m;
    std::vector<SideStream> all;
    const SideStream* get(hipStream_t main) {
        std::lock_guard<std::mutex> lock(m);
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return nullptr;
        for (auto& e : all) if (e.dev == dev && e.main == main) return &e;
        SideStream e{dev, main, nullptr, nullptr, nullptr};
        if (hipStreamCreateWithFlags(&e.side, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&e.in, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&e.done, hipEventDisableTiming) != hipSuccess) return nullptr;
        all.reserve(64);
        if (all.size() >= 64) return nullptr;              // pointers into `all` stay valid: it never reallocates
        all.push_back(e);
        return &all.back();
    }
} g_side;

// modes[kx + MBOX][ky + MBOX][kz] (+)= sum over the nx planes x0 .. x0 + nx - 1 (rows of n floats, n rows per plane) of
// f(x, y, z) e^{-2 pi i (kx x + ky y + kz z) / n}.  work: nx * n * (MBOX + 1) + nx * 13 * 7 + 64 * 1183 double2.
static int lowk_modes(const float* planes, const float* rec, int window, int n, int x0, int nx, int accumulate,
                      double2* modes, double2* work, hipStream_t s, bool z_done = false) {
    double2* lowz = work;                                                  // [x][y][kz]
    double2* lowy = lowz + (size_t)nx * n * (MBOX + 1);                    // [x][ky][kz]
    double2* parts = lowy + (size_t)nx * (2 * MBOX + 1) * (MBOX + 1);      // [part][kx][ky][kz]
    const size_t nrows = (size_t)nx * n;
    const unsigned blocks = (unsigned)std::min<size_t>((nrows / 2 + 3) / 4, 256 * 12);  // waves walk many row pairs each
    auto z = [&](auto nj) {
        constexpr int NJ = decltype(nj)::value;
        if (rec == nullptr) lowk_z_kernel<NJ, 0><<<blocks, 256, 0, s>>>(planes, rec, n, nrows, lowz);
        else if (window == AST_WIN_CIC) lowk_z_kernel<NJ, 2><<<blocks, 256, 0, s>>>(planes, rec, n, nrows, lowz);
        else lowk_z_kernel<NJ, 3><<<blocks, 256, 0, s>>>(planes, rec, n, nrows, lowz);
    };
    if (z_done) {}                                       // the FFT's z pass has left the z sums in `work`
    else if (n == 2048) z(std::integral_constant<int, 32>{});
    else if (n == 1024) z(std::integral_constant<int, 16>{});
    else if (n == 512) z(std::integral_constant<int, 8>{});
    else z(std::integral_constant<int, 4>{});
    const size_t lds = (size_t)n * sizeof(double2);
    lowk_axis_kernel<<<dim3((unsigned)nx, 1), 256, lds, s>>>(lowz, n, n, 0, MBOX + 1, lowy);
    const int inner = (2 * MBOX + 1) * (MBOX + 1);
    lowk_axis_kernel<<<dim3(1, LOWK_XPARTS), 256, lds, s>>>(lowy, n, nx, x0, inner, parts);
    lowk_parts_reduce_kernel<<<(unsigned)((LOWK_MODES + 255) / 256), 256, 0, s>>>(parts, LOWK_XPARTS, (int)LOWK_MODES, accumulate, modes);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

static int power_3d_impl(const void* grid, void* scratch, size_t scratch_bytes, int dtype, size_t n, double boxsize,
                         double mean, double* psum, const void* rec, int window, int lowk, int binning, void* stream) {
    AST_CHECK_ARG(grid != nullptr && scratch != nullptr && psum != nullptr && boxsize > 0.0);
    AST_CHECK_ARG(lowk == 0 || lowk == 1);
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    const double kf_rule = binning == AST_BIN_FLOAT64 ? 2.0 * M_PI / boxsize : 0.0;
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    AST_CHECK_ARG(scratch_bytes >= ast_fft_tile_power_scratch_bytes(n));
    const size_t nz = n / 2 + 1, nzp = (nz + 15) / 16 * 16, tiles = (nz + 15) / 16;
    float2* spec = (float2*)scratch;
    float2* disc = (float2*)((char*)scratch + n * n * nzp * sizeof(float2));
    double* partial = (double*)((char*)disc + power_disc_bytes(n));
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_power_3d: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    double* lowk_sums = (double*)((char*)scratch + power_core_bytes(n)) - 64;
    const SideStream* side = nullptr;
    double2* modes = (double2*)((char*)scratch + (power_core_bytes(n) + 255) / 256 * 256);
    double2* work = (double2*)((char*)modes + (LOWK_MODES * sizeof(double2) + 255) / 256 * 256);
    // the z pass forms the low-k z sums from the samples it holds anyway (AST_LOWK_SEPARATE: lowk_z_kernel instead)
    const bool z_fused = lowk && !getenv("AST_LOWK_SEPARATE");
    auto lowk_rest = [&]() -> int {
        // the modes |m_i| <= MBOX as DFT sums in double, on the side stream, in their own part of the scratch
        AST_CHECK_HIP(hipEventRecord(side->in, s));                  // the grid (or the z sums) and the scratch are ready
        AST_CHECK_HIP(hipStreamWaitEvent(side->side, side->in, 0));
        {
            AST_PROF("fft_tile.lowk", side->side);
            int rc = lowk_modes((const float*)grid, (const float*)rec, window, (int)n, 0, (int)n, 0, modes, work, side->side, z_fused);
            if (rc != AST_OK) return rc;
            const double ng = (double)n * (double)n * (double)n;
            lowk_shell_kernel<<<1, 256, 0, side->side>>>(modes, boxsize * boxsize * boxsize / (ng * ng), kf_rule, lowk_sums);
            AST_CHECK_LAUNCH();
        }
        AST_CHECK_HIP(hipEventRecord(side->done, side->side));
        return AST_OK;
    };
    if (lowk) {
        side = g_side.get(s);
        if (!side) { ast::set_error("ast_fft_tile_power_3d: side stream creation failed"); return AST_ERR_HIP; }
        if (!z_fused) { const int rc = lowk_rest(); if (rc != AST_OK) return rc; }
    }
    int rc = rows_r2c_impl(grid, spec, dtype, n, n * n, n, nzp, 1.0, mean, stream, rec, window,
                           z_fused ? reinterpret_cast<double*>(work) : nullptr);                       // z
    if (rc != AST_OK) return rc;
    if (z_fused) { rc = lowk_rest(); if (rc != AST_OK) return rc; }
    // both strided passes leave out what FFTPower drops: rows / tiles with k_y^2 + k_z0^2 > (n/2)^2 (AST_FFT_NO_PRUNE: A/B runs)
    const long long prune2 = getenv("AST_FFT_NO_PRUNE") ? 0 : (long long)(n / 2) * (long long)(n / 2);
    const double inv_ng = 1.0 / ((double)n * (double)n * (double)n);
    // AST_FFT_DISC=1 (A/B runs): the k_y pass stores in the disc layout (one part) and the last pass reads it from there;
    // default: in place on the pitched rows, rows / tiles outside the disc skipped
    const char* disc_env = getenv("AST_FFT_DISC");
    const bool use_disc = prune2 != 0 && disc_env != nullptr && (disc_env[0] == '1' || disc_env[0] == '2');
    const unsigned xmajor = use_disc && disc_env[0] == '2' ? (unsigned)n : 0u;      // 2: x-major tiles (experiment)
    if (use_disc) {
        const DiscLayout::Dev* dv = nullptr;
        const DiscLayout* L = g_disc.get(n, 1, true, &dv);
        if (!L) { ast::set_error("ast_fft_tile_power_3d: disc table allocation failed"); return AST_ERR_HIP; }
        {
            AST_PROF("fft_tile.c2c", s);
            PackDst pk;
            pk.nbatch = (unsigned)n;
            pk.disc = dv->tab;
            pk.self_out = disc;
            pk.self_part = 0;
            pk.xmajor = xmajor;
            if (n == 1024) rc = launch_c2c_pack<32, 32, 16, 2>(spec, tw, nzp, nz, n, n * nzp, 1.0f, pk, s);
            else if (n == 512) rc = launch_c2c_pack<16, 32, 16, 2>(spec, tw, nzp, nz, n, n * nzp, 1.0f, pk, s);
            else rc = launch_c2c_pack<16, 16, 16, 2>(spec, tw, nzp, nz, n, n * nzp, 1.0f, pk, s);       // y, per x-plane
            if (rc != AST_OK) return rc;
        }
        AST_PROF("fft_tile.c2c_power", s);
        const unsigned* edge_fall = nullptr;
        if (kf_rule != 0.0) {
            edge_fall = g_edge.get(n, boxsize, s);
            if (!edge_fall) { ast::set_error("ast_fft_tile_power_3d: edge table allocation failed"); return AST_ERR_HIP; }
        }
        PackDst src;
        src.disc = dv->tab;
        src.gk = dv->gk;
        src.nbatch = (unsigned)n;
        src.xmajor = xmajor;
        rc = dispatch_c2c<true>(n, disc, tw, xmajor ? 16 : L->S[0], nz, n, 16, (float)inv_ng, partial, s, edge_fall, 0, 0, src);   // x + binning
        if (rc != AST_OK) return rc;
    } else {
        {
            AST_PROF("fft_tile.c2c", s);
            rc = dispatch_c2c<false>(n, spec, tw, nzp, nz, n, n * nzp, 1.0f, nullptr, s, nullptr, 0, prune2);       // y, per x-plane
            if (rc != AST_OK) return rc;
        }
        AST_PROF("fft_tile.c2c_power", s);
        const unsigned* edge_fall = nullptr;
        if (kf_rule != 0.0) {
            edge_fall = g_edge.get(n, boxsize, s);
            if (!edge_fall) { ast::set_error("ast_fft_tile_power_3d: edge table allocation failed"); return AST_ERR_HIP; }
        }
        rc = dispatch_c2c<true>(n, spec, tw, n * nzp, nz, n, nzp, (float)inv_ng, partial, s, edge_fall, 0, prune2);   // x + binning
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft_tile.shell_reduce", s);
    const int nb = (int)(n / 2 - 1);
    double* partial2 = (double*)scratch;                   // the spectrum scratch is dead after the x pass
    shell_partials_stage1_kernel<<<REDUCE_ROWS, 256, 0, s>>>(partial, n * tiles, nb, partial2);
    shell_partials_stage2_kernel<<<(nb + 7) / 8, 256, 0, s>>>(partial2, nb, boxsize * boxsize * boxsize, lowk ? MLOW : 0, psum);
    if (lowk) {
        AST_CHECK_HIP(hipStreamWaitEvent(s, side->done, 0));
        lowk_patch_kernel<<<1, 64, 0, s>>>(lowk_sums, MLOW, psum);
    }
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// The low-k channel as separate calls (slab-decomposed grids: every rank adds its planes, the modes are all-reduced,
// then the shell sums are taken).
extern "C" size_t ast_lowk_work_bytes(size_t n, size_t nx) {
    return (nx * n * (MBOX + 1) + nx * (2 * MBOX + 1) * (MBOX + 1) + (size_t)LOWK_XPARTS * LOWK_MODES) * sizeof(double2);
}
extern "C" int ast_lowk_mode_count(void) { return (int)LOWK_MODES; }
extern "C" int ast_lowk_shell_count(void) { return MLOW; }

extern "C" int ast_lowk_modes(const void* planes, int dtype, size_t n, size_t x0, size_t nx, int accumulate, void* modes,
                              void* work, size_t work_bytes, void* stream) {
    AST_CHECK_ARG(planes && modes && work && nx >= 1 && x0 + nx <= n && (nx * n) % 4 == 0);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n) || (dtype == AST_F32 && n == 2048));      // (2048: the passes of lens_fft.hip's fp32 section)
    AST_CHECK_ARG(work_bytes >= ast_lowk_work_bytes(n, nx));
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("fft_tile.lowk", s);
    return lowk_modes((const float*)planes, nullptr, 0, (int)n, (int)x0, (int)nx, accumulate, (double2*)modes, (double2*)work, s);
}

// ast_lowk_modes for planes whose z sums are already at the start of work_d ([plane][y][kz], nx * n * 7 complex128,
// written by ast_fft_tile_rows_r2c_lowz plane range by plane range): the y and x sums only.
extern "C" int ast_lowk_modes_from_z(size_t n, size_t x0, size_t nx, int accumulate, void* modes, void* work, size_t work_bytes,
                                     void* stream) {
    AST_CHECK_ARG(modes && work && nx >= 1 && x0 + nx <= n);
    AST_CHECK_ARG(ast_fft_tile_supported(AST_F32, n));
    AST_CHECK_ARG(work_bytes >= ast_lowk_work_bytes(n, nx));
    hipStream_t s = ast::as_stream(stream);
    AST_PROF("fft_tile.lowk", s);
    return lowk_modes(nullptr, nullptr, 0, (int)n, (int)x0, (int)nx, accumulate, (double2*)modes, (double2*)work, s, true);
}

extern "C" int ast_lowk_shell_sums(const void* modes, size_t n, double boxsize, int binning, double* sums, void* stream) {
    AST_CHECK_ARG(modes && sums && boxsize > 0.0 && n >= 16);
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    const double ng = (double)n * (double)n * (double)n;
    lowk_shell_kernel<<<1, 256, 0, ast::as_stream(stream)>>>((const double2*)modes, boxsize * boxsize * boxsize / (ng * ng),
                                                          binning == AST_BIN_FLOAT64 ? 2.0 * M_PI / boxsize : 0.0, sums);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// The unnormalised inverse of ast_fft_tile_r2c_3d's layout, real_out[x] = sum_k spec_k e^{+ikx}, for an (n, n, n/2+1)
// half spectrum, in three tile passes (x, y, z).  spec_d is NOT modified; work_d (same size as spec_d) is.  With
// m_hi > m_lo >= 0 only the modes with m_lo <= |m| < m_hi enter (the shell filter of the bispectrum estimator, fused
// into the first pass's loads); m_hi = 0: all modes.
extern "C" int ast_fft_tile_c2r_3d(const void* spec, void* work, void* out, int dtype, size_t n, int m_lo, int m_hi,
                                   double scale, void* stream) {
    AST_CHECK_ARG(spec != nullptr && work != nullptr && out != nullptr && spec != work && work != out);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    AST_CHECK_ARG(m_lo >= 0 && (m_hi == 0 || m_hi > m_lo));
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_c2r_3d: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    const size_t nz = n / 2 + 1;
    ShellMask mask{(const float2*)spec, (long long)m_lo * m_lo, (long long)m_hi * m_hi};
    {
        AST_PROF("fft_tile.c2c_inv", s);
        // x: rows k_x (element stride n * nz), batch k_y (stride nz), columns k_z; masked loads from spec, stores to work
        int rc = dispatch_c2c_inv(n, (float2*)work, tw, n * nz, nz, n, nz, 1.0f, mask, s);
        if (rc != AST_OK) return rc;
        // y: per x plane, rows k_y (stride nz), in place
        rc = dispatch_c2c_inv(n, (float2*)work, tw, nz, nz, n, n * nz, 1.0f, ShellMask{nullptr, 0, mask.hi2, 0, 1}, s);
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft_tile.rows_c2r", s);
    const int kmax = m_hi > 0 && (size_t)m_hi < nz ? m_hi : (int)nz;
    if (n == 1024) return launch_c2r<16, 32, 16>((const float2*)work, (float*)out, tw, n * n, nz, n, (float)scale, kmax, s);
    if (n == 512) return launch_c2r<16, 16, 16>((const float2*)work, (float*)out, tw, n * n, nz, n, (float)scale, kmax, s);
    return launch_c2r<8, 16, 16>((const float2*)work, (float*)out, tw, n * n, nz, n, (float)scale, kmax, s);
}

// ast_fft_tile_c2r_3d for up to SHELL_BATCH shells of ONE spectrum in three launches (x, y, z with blockIdx.y = shell):
// works[i] / outs[i] are shell i's scratch spectrum and real output (host arrays of device pointers), m_lo[i] < m_hi[i] its
// radii.  Same arithmetic per shell as the single call (bit-identical outputs).  work_pitch: row pitch of the scratch spectra
// in complex elements (0: n / 2 + 1).  A multiple of 16 makes the 128-byte row pieces of the x / y passes' column tiles whole
// lines - at the natural pitch every piece straddles two, both shared with the neighbouring tiles.
extern "C" int ast_fft_tile_c2r_3d_batch(const void* spec, void* const* works, void* const* outs, int dtype, size_t n,
                                         const int* m_lo, const int* m_hi, int count, double scale, int passes, size_t work_pitch,
                                         void* stream) {
    AST_CHECK_ARG(spec != nullptr && works != nullptr && outs != nullptr && m_lo != nullptr && m_hi != nullptr);
    AST_CHECK_ARG(passes >= 1 && passes <= 3);
    AST_CHECK_ARG(work_pitch == 0 || work_pitch >= n / 2 + 1);
    AST_CHECK_ARG(count >= 1 && count <= SHELL_BATCH);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_c2r_3d_batch: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    const size_t nz = n / 2 + 1, wp = work_pitch ? work_pitch : nz;      // row pitch of the scratch spectra
    ShellBatch sb;
    C2RBatch cb;
    sb.count = cb.count = count;
    for (int i = 0; i < SHELL_BATCH; ++i) {
        const int j = i < count ? i : count - 1;                  // (unused slots repeat the last shell)
        AST_CHECK_ARG(works[j] != nullptr && works[j] != spec);
        AST_CHECK_ARG(!(passes & 2) || (outs[j] != nullptr && works[j] != outs[j]));       // (no z pass: no output needed)
        AST_CHECK_ARG(m_lo[j] >= 0 && m_hi[j] > m_lo[j]);
        sb.work[i] = (float2*)works[j];
        sb.lo2[i] = (long long)m_lo[j] * m_lo[j];
        sb.hi2[i] = (long long)m_hi[j] * m_hi[j];
        cb.in[i] = (const float2*)works[j];
        cb.out[i] = (float*)outs[j];
        cb.kmax[i] = (size_t)m_hi[j] < nz ? m_hi[j] : (int)nz;
    }
    for (int i = 0; i < count; ++i)
        for (int k = 0; k < i; ++k) AST_CHECK_ARG(works[i] != works[k] && (!(passes & 2) || outs[i] != outs[k]));
    if (passes & 1) {
        AST_PROF("fft_tile.c2c_inv", s);
        ShellMask mx{(const float2*)spec, 0, 1, 0, 0, n * nz, nz};      // radii come from the batch; hi2 > 0 switches the pruning on
        int rc = dispatch_c2c_inv(n, sb.work[0], tw, n * wp, nz, n, wp, 1.0f, mx, s, sb);
        if (rc != AST_OK) return rc;
        ShellMask my{nullptr, 0, 1, 0, 1};
        rc = dispatch_c2c_inv(n, sb.work[0], tw, wp, nz, n, n * wp, 1.0f, my, s, sb);
        if (rc != AST_OK) return rc;
    }
    if (!(passes & 2)) return AST_OK;
    AST_PROF("fft_tile.rows_c2r", s);
    if (n == 1024) return launch_c2r<16, 32, 16>(cb.in[0], cb.out[0], tw, n * n, wp, n, (float)scale, cb.kmax[0], s, cb);
    if (n == 512) return launch_c2r<16, 16, 16>(cb.in[0], cb.out[0], tw, n * n, wp, n, (float)scale, cb.kmax[0], s, cb);
    return launch_c2r<8, 16, 16>(cb.in[0], cb.out[0], tw, n * n, wp, n, (float)scale, cb.kmax[0], s, cb);
}

// The last pass of a slab-decomposed transform fused with the shell binning: `block_d` is a rank's (n, nloc, pitch)
// block of the spectrum after the all-to-all (all k_x, k_y = ky0 .. ky0 + nloc - 1, half k_z; row pitch `pitch` >= n/2+1
// complex); the axis-0 pass runs over it and, instead of storing delta_k, adds w |delta_k|^2 of its modes to
// psum_d (+=, L^3 sum w |delta_k|^2 with delta_k scaled by `scale`; the block's contents afterwards are undefined).
// scratch_d: ast_fft_tile_block_power_scratch_bytes(n, nloc) bytes.  first_bin: shells below it are skipped (they
// come from the low-k channel).
extern "C" size_t ast_fft_tile_block_power_scratch_bytes(size_t n, size_t nloc) {
    const size_t tiles = (n / 2 + 1 + 15) / 16, nb = n / 2 - 1;
    return (nloc * tiles + REDUCE_ROWS) * nb * sizeof(double);
}

extern "C" int ast_fft_tile_block_power(void* block, void* scratch, size_t scratch_bytes, int dtype, size_t n, size_t nloc,
                                        size_t ky0, size_t pitch, double scale, double boxsize, int first_bin, int binning,
                                        double* psum, void* stream) {
    AST_CHECK_ARG(block != nullptr && scratch != nullptr && psum != nullptr && boxsize > 0.0);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n));
    AST_CHECK_ARG(nloc >= 1 && ky0 + nloc <= n && pitch >= n / 2 + 1 && first_bin >= 0);
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    AST_CHECK_ARG(scratch_bytes >= ast_fft_tile_block_power_scratch_bytes(n, nloc));
    const size_t nz = n / 2 + 1, tiles = (nz + 15) / 16;
    const int nb = (int)(n / 2 - 1);
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_block_power: twiddle table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    const unsigned* edge_fall = nullptr;
    if (binning == AST_BIN_FLOAT64) {
        edge_fall = g_edge.get(n, boxsize, s);
        if (!edge_fall) { ast::set_error("ast_fft_tile_block_power: edge table allocation failed"); return AST_ERR_HIP; }
        edge_fall += ky0 * tiles * (n == 256 ? 256 : 512);          // the table is [k_y][tile][thread]
    }
    double* partial = (double*)scratch;
    double* partial2 = partial + nloc * tiles * nb;
    {
        AST_PROF("fft_tile.c2c_power", s);
        int rc = dispatch_c2c<true>(n, (float2*)block, tw, nloc * pitch, nz, nloc, pitch, (float)scale, partial, s, edge_fall, (int)ky0);
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft_tile.shell_reduce", s);
    shell_partials_stage1_kernel<<<REDUCE_ROWS, 256, 0, s>>>(partial, nloc * tiles, nb, partial2);
    shell_partials_stage2_kernel<<<(nb + 7) / 8, 256, 0, s>>>(partial2, nb, boxsize * boxsize * boxsize, first_bin, psum);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ------------------------------------------------------------------ disc layout: C-ABI
// plane_elems[parts]: complex elements of one plane of each part; block_part[n / r1]: owner of every row block (may be NULL);
// r1_out: rows per block.  Pure host arithmetic (no GPU needed).
extern "C" int ast_fft_tile_disc_layout(size_t n, int parts, unsigned* plane_elems, unsigned char* block_part, int* r1_out) {
    AST_CHECK_ARG(disc_geometry_ok(n, parts) && plane_elems != nullptr);
    const DiscLayout* L = g_disc.get(n, parts, false, nullptr);
    if (!L) { ast::set_error("ast_fft_tile_disc_layout: table construction failed"); return AST_ERR_HIP; }
    for (int r = 0; r < parts; ++r) plane_elems[r] = L->S[r];
    if (block_part) for (unsigned k = 0; k < L->nblk; ++k) block_part[k] = L->part_of[k];
    if (r1_out) *r1_out = L->r1;
    return AST_OK;
}

// The table itself, [tile][row block] x (offb, S, cumS, lohi) as 4 x int32 (tests; host memory).
extern "C" int ast_fft_tile_disc_table(size_t n, int parts, int* out, size_t out_ints) {
    AST_CHECK_ARG(disc_geometry_ok(n, parts) && out != nullptr);
    const DiscLayout* L = g_disc.get(n, parts, false, nullptr);
    if (!L) { ast::set_error("ast_fft_tile_disc_table: table construction failed"); return AST_ERR_HIP; }
    AST_CHECK_ARG(out_ints >= L->tab.size() * 4);
    memcpy(out, L->tab.data(), L->tab.size() * sizeof(DiscEntry));
    return AST_OK;
}

// The k_y pass of `nplanes` local planes (planes_d: (nplanes, n, pitch) complex, the z pass's output; left intact) storing in
// the disc layout: part q's rows of plane b go to packed_d + nplanes * cumS[q] + b * S[q] (what is sent to rank q: nplanes
// contiguous planes of S[q] elements), the part `self_part` to self_out_d + b * S[self_part] instead (the rank's own piece,
// straight into its receive block; self_part < 0: none).  Rows outside the Nyquist disc are not stored anywhere.
extern "C" int ast_fft_tile_c2c_disc(const void* planes, void* packed, int dtype, size_t n, size_t pitch, size_t nplanes,
                                     int parts, int self_part, void* self_out, double scale, void* stream) {
    AST_CHECK_ARG(planes != nullptr && planes != packed && planes != self_out && nplanes >= 1 && pitch >= n / 2 + 1);
    AST_CHECK_ARG(dtype == AST_F32 && disc_geometry_ok(n, parts));
    AST_CHECK_ARG((self_out == nullptr) == (self_part < 0) && self_part < parts);
    AST_CHECK_ARG(packed != nullptr || (self_out != nullptr && parts == 1));
    AST_CHECK_ARG(pitch < (1u << 24));
    const DiscLayout::Dev* dv = nullptr;
    const DiscLayout* L = g_disc.get(n, parts, true, &dv);
    const float2* tw = g_tw.get((int)n);
    if (!L || !tw) { ast::set_error("ast_fft_tile_c2c_disc: table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    PackDst pack;
    pack.out = (float2*)packed;
    pack.nbatch = (unsigned)nplanes;
    pack.disc = dv->tab;
    if (self_out) { pack.self_out = (float2*)self_out; pack.self_part = (unsigned)self_part; }
    AST_PROF("fft_tile.c2c", s);
    float2* d = (float2*)const_cast<void*>(planes);
    const size_t ncols = n / 2 + 1;
    if (n == 1024) return launch_c2c_pack<32, 32, 16, 2>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
    if (n == 512) return launch_c2c_pack<16, 32, 16, 2>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
    return launch_c2c_pack<16, 16, 16, 2>(d, tw, pitch, ncols, nplanes, n * pitch, (float)scale, pack, s);
}

// The last pass of a slab-decomposed transform over a part's block in the disc layout ((n planes, S[part]) complex, what
// ast_fft_tile_c2c_disc wrote / the transpose delivered), fused with the shell binning: psum_d += L^3 sum w |delta_k|^2 of the
// part's modes (delta_k scaled by `scale`).  scratch_d: ast_fft_tile_disc_power_scratch_bytes(n, parts).
extern "C" size_t ast_fft_tile_disc_power_scratch_bytes(size_t n, int parts) {
    if (!disc_geometry_ok(n, parts)) return 0;
    return ast_fft_tile_block_power_scratch_bytes(n, n / (size_t)parts);
}

extern "C" int ast_fft_tile_disc_block_power(void* block, void* scratch, size_t scratch_bytes, int dtype, size_t n, int parts,
                                             int part, double scale, double boxsize, int first_bin, int binning, double* psum,
                                             void* stream) {
    AST_CHECK_ARG(block != nullptr && scratch != nullptr && psum != nullptr && boxsize > 0.0 && first_bin >= 0);
    AST_CHECK_ARG(dtype == AST_F32 && disc_geometry_ok(n, parts) && part >= 0 && part < parts);
    AST_CHECK_ARG(binning == AST_BIN_INTEGER || binning == AST_BIN_FLOAT64);
    AST_CHECK_ARG(scratch_bytes >= ast_fft_tile_disc_power_scratch_bytes(n, parts));
    const size_t nz = n / 2 + 1, tiles = (nz + 15) / 16, nloc = n / (size_t)parts;
    const int nb = (int)(n / 2 - 1);
    const DiscLayout::Dev* dv = nullptr;
    const DiscLayout* L = g_disc.get(n, parts, true, &dv);
    const float2* tw = g_tw.get((int)n);
    if (!L || !tw) { ast::set_error("ast_fft_tile_disc_block_power: table allocation failed"); return AST_ERR_HIP; }
    hipStream_t s = ast::as_stream(stream);
    const unsigned* edge_fall = nullptr;
    if (binning == AST_BIN_FLOAT64) {
        edge_fall = g_edge.get(n, boxsize, s);                       // [k_y][tile][thread]: indexed by the lattice row
        if (!edge_fall) { ast::set_error("ast_fft_tile_disc_block_power: edge table allocation failed"); return AST_ERR_HIP; }
    }
    PackDst src;
    src.disc = dv->tab;
    src.gk = dv->gk;
    src.part = (unsigned)part;
    src.nbatch = (unsigned)nloc;
    double* partial = (double*)scratch;
    double* partial2 = partial + nloc * tiles * nb;
    {
        AST_PROF("fft_tile.c2c_power", s);
        // (elem_stride / batch_stride only feed the host's range check here: the kernel takes both from the table)
        int rc = dispatch_c2c<true>(n, (float2*)block, tw, L->S[part], nz, nloc, 16, (float)scale, partial, s, edge_fall, 0, 0, src);
        if (rc != AST_OK) return rc;
    }
    AST_PROF("fft_tile.shell_reduce", s);
    shell_partials_stage1_kernel<<<REDUCE_ROWS, 256, 0, s>>>(partial, nloc * tiles, nb, partial2);
    shell_partials_stage2_kernel<<<(nb + 7) / 8, 256, 0, s>>>(partial2, nb, boxsize * boxsize * boxsize, first_bin, psum);
    AST_CHECK_LAUNCH();
    return AST_OK;
}

// ------------------------------------------------------------------ bispectrum: z passes of all shells + triangle sums in one kernel
// works[s]: shell s's scratch spectrum after the masked x and y inverse passes (ast_fft_tile_c2r_3d_batch with passes = 1),
// (n, n, work_pitch) complex64; m_hi[s] its outer radius (k_z < m_hi is all the passes wrote).  out_d[t] = sum over the n^3 cells of
// f_a f_b f_c for triangle t = (a, b, c) (shell slots, device int array tri_d (ntri, 3)), f_s = scale * C2R_z(works[s]) - what
// ast_fft_tile_c2r_3d_batch(passes = 2) followed by ast_triple_product_sums computes, without the real cubes ever reaching HBM.
// nshells <= 32, ntri <= 512 (1024 at n = 1024).  scratch_d: ast_fft_tile_c2r_triangles_scratch_bytes().
constexpr int TRI_GRID_MAX = 2048;
extern "C" size_t ast_fft_tile_c2r_triangles_scratch_bytes(void) { return (size_t)TRI_GRID_MAX * 1024 * sizeof(double); }

extern "C" int ast_fft_tile_c2r_triangles(void* const* works, const int* m_hi, int nshells, int dtype, size_t n, size_t work_pitch,
                                          double scale, const int* tri, int ntri, void* scratch, double* out, void* stream) {
    AST_CHECK_ARG(works != nullptr && m_hi != nullptr && tri != nullptr && scratch != nullptr && out != nullptr);
    AST_CHECK_ARG(ast_fft_tile_supported(dtype, n) && nshells >= 1 && nshells <= TRI_ROWS);
    const size_t nz = n / 2 + 1, wp = work_pitch ? work_pitch : nz;
    AST_CHECK_ARG(wp >= nz);
    const int nt = TRI_ROWS * (n == 1024 ? 32 : 16);
    AST_CHECK_ARG(ntri >= 1 && ntri <= nt);
    const float2* tw = g_tw.get((int)n);
    if (!tw) { ast::set_error("ast_fft_tile_c2r_triangles: twiddle table allocation failed"); return AST_ERR_HIP; }
    TriShells sh;
    sh.count = nshells;
    for (int i = 0; i < TRI_ROWS; ++i) {
        const int j = i < nshells ? i : nshells - 1;
        AST_CHECK_ARG(works[j] != nullptr && m_hi[j] > 0);
        sh.work[i] = (const float2*)works[j];
        sh.kmax[i] = (size_t)m_hi[j] < nz ? m_hi[j] : (int)nz;
    }
    hipStream_t s = ast::as_stream(stream);
    const int parts = nt / ntri;
    const size_t nrows = n * n;
    AST_PROF("fft_tile.c2r_triangles", s);
    auto go = [&](auto r1, auto r2) -> int {
        constexpr int R1 = decltype(r1)::value, R2 = decltype(r2)::value, M = R1 * R2, NT = TRI_ROWS * R2;
        constexpr int BUF = TRI_ROWS * (R1 * (R2 + 1) > M + 1 ? R1 * (R2 + 1) : M + 1);
        const size_t lds = (size_t)(BUF + 2 * M) * sizeof(float2);
        static ast::PerDeviceOnce attr_once;
        if (attr_once.need()) {
            AST_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&rows_c2r_triangles_kernel<R1, R2>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_once.mark();
        }
        // persistent workgroups: as many as stay resident (LDS: two per CU up to n = 512, one at 1024), a few rounds each
        const size_t resident = 256 * (lds <= 80 * 1024 ? 2 : 1);
        const int blocks = (int)std::min<size_t>(std::min<size_t>(nrows, resident * 4), (size_t)TRI_GRID_MAX);
        rows_c2r_triangles_kernel<R1, R2><<<blocks, NT, lds, s>>>(sh, tw, nrows, wp, tri, ntri, parts, (double*)scratch);
        AST_CHECK_LAUNCH();
        // every stored value is x / (2 scale'): the kernel keeps z, x = 2 z; three factors of 2 scale
        const double f = 2.0 * scale;
        triangles_reduce_kernel<<<ntri, 256, 0, s>>>((const double*)scratch, blocks, NT, ntri, parts, f * f * f, out);
        AST_CHECK_LAUNCH();
        return AST_OK;
    };
    if (n == 1024) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 32>{});
    if (n == 512) return go(std::integral_constant<int, 16>{}, std::integral_constant<int, 16>{});
    return go(std::integral_constant<int, 8>{}, std::integral_constant<int, 16>{});
}
