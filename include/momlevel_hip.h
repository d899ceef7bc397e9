/*
 * momlevel_hip.h -- C ABI of libmomlevel_hip.so, the MI355X (gfx950) implementation
 * of momlevel's steric hot path.
 *
 * The reference (jkrasting/momlevel) is pure Python: it has no FFI.  Its boundary
 * for this path is a set of Python call signatures over numpy/xarray arrays.  Each
 * entry point below names the reference function(s) whose array arithmetic it
 * replaces (file:line into the reference tree); momlevel_amd/ binds them with
 * ctypes (INTEGRATION.md shows the stub a momlevel maintainer would add).
 *
 * Conventions
 *  - Every function returns int: 0 = OK, <0 = argument error (MLX_E_*), >0 = a
 *    hipError_t.  Nothing throws.  mlx_last_error() gives the text for the last
 *    non-zero status returned on the calling thread.
 *  - The caller owns every buffer.  All array pointers are DEVICE pointers (HBM),
 *    C-contiguous, x fastest: 4-D (time, z_l, yh, xh), 3-D (z_l, yh, xh), 2-D (yh, xh).
 *    "plane" is ny*nx.  Strides are in ELEMENTS.  The library allocates nothing:
 *    scratch is a caller-provided workspace sized by the *_workspace_bytes() query.
 *  - Kernels are enqueued asynchronously on the caller's hipStream_t (passed as
 *    void*; NULL = the default stream).  No call synchronises.  No global mutable
 *    state: calls on different streams may run from different host threads.
 *  - NaN marks land / below-bottom cells.  Reductions are skipna (a NaN term
 *    counts as 0; an all-NaN reduction gives 0.0), as xarray's default .sum().
 *  - Pointwise outputs (rho, its derivatives, delta_rho, dz, local eta) are
 *    bit-identical to the reference's numpy evaluation on finite inputs: the
 *    device code keeps the reference's operator order with FMA contraction off
 *    (unless the caller opts into MLX_FLAG_FMA).
 *    Reductions over (z,y,x) differ from numpy's pairwise order at the 1e-15 level
 *    and are deterministic (fixed order, no float atomics).
 */
#ifndef MOMLEVEL_HIP_H
#define MOMLEVEL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLX_ABI_VERSION 8 /* 2: mlx_eos_map takes flags; mlx_steric_global_decomp, mlx_steric_local_decomp,
                             mlx_stream_probe;
                             MLX_FLAG_FMA, MLX_FLAG_TCHUNK; MLX_P_FULL4D in K1/K2
                             3: mlx_build_kind; a NULL p (linear EOS) requires p_mode MLX_P_SCALAR
                             4: mlx_eos_map_promote (MLX_KIND_*); MLX_DTYPE_T32_S64 / _T64_S32 in K1/K2
                             5: mlx_stream_probe_mix, mlx_valu_probe, mlx_last_kernel
                             6: mlx_host_copy; mlx_stratification, mlx_adjust_negative_n2,
                                mlx_wave_speed_where_time0
                             7: mlx_host_copy_masked
                             8: MLX_FUNC_DENSITY_REF in mlx_eos_map_promote */

/* argument-error codes (negative) */
#define MLX_E_NULL     (-1) /* a required pointer is NULL                      */
#define MLX_E_SHAPE    (-2) /* a dimension is <= 0 or too large for the grid   */
#define MLX_E_ENUM     (-3) /* eos / func / p_mode / dtype value not recognised */
#define MLX_E_WORKSPACE (-4) /* workspace too small or misaligned              */
#define MLX_E_ALIGN    (-5) /* pointer not aligned to its element size         */

/* equation of state: momlevel.eos.<name>  (src/momlevel/util.py:227-249) */
#define MLX_EOS_WRIGHT 0 /* src/momlevel/eos/wright.py */
#define MLX_EOS_LINEAR 1 /* src/momlevel/eos/linear.py */

/* EOS function: the attribute picked from the eos module */
#define MLX_FUNC_DENSITY    0 /* eos/wright.py:23-50   */
#define MLX_FUNC_DRHO_DTEMP 1 /* eos/wright.py:53-85   */
#define MLX_FUNC_DRHO_DSAL  2 /* eos/wright.py:88-119  */
#define MLX_FUNC_ALPHA      3 /* eos/wright.py:122-142 */
#define MLX_FUNC_BETA       4 /* eos/wright.py:145-165 */
#define MLX_FUNC_IBH        5 /* dynamic.py:34-36, via mlx_inverse_barometer only */
#define MLX_FUNC_DENSITY_REF 6 /* eos/linear.py:55-56 with rho_ref given; mlx_eos_map_promote only */

/* how the pressure argument is laid out */
#define MLX_P_SCALAR  0 /* p[0] for every cell (calc_pdens, scalar calls)            */
#define MLX_P_ZPROF   1 /* p[nz]: the steric path, pres = z_l*1e4 + patm              */
#define MLX_P_FULL3D  2 /* p[nz*plane]: patm given as a (yh,xh) DataArray             */
#define MLX_P_FULL4D  3 /* p[nt*nz*plane]: patm given as a (time,yh,xh) DataArray        */

/* dtype of the streamed theta/S fields */
#define MLX_DTYPE_F64 0
#define MLX_DTYPE_F32 1 /* numpy mixed precision of the reference: al0,p0,lam rounded
                           in float32, the rest in float64        */
#define MLX_DTYPE_F32_UPCAST 2 /* float32 storage, upcast to float64 before any math  */
/* theta and salinity of DIFFERENT dtypes (a dataset whose variables were written with different
 * precisions).  numpy then evaluates every sub-expression of eos/wright.py:44-46 that involves one
 * field only in that field's precision and joins the two in float64; the steric entry points
 * (mlx_steric_global*, mlx_steric_local*) reproduce that bit for bit, in exact arithmetic only
 * (MLX_FLAG_FMA is refused).  T / T0 point at the first type, S / S0 at the second; strides are in
 * elements of each field's own type.  The pointwise maps (mlx_eos_map, mlx_inverse_barometer) refuse
 * these two: mlx_eos_map_promote covers every combination there. */
#define MLX_DTYPE_T32_S64 3 /* theta float32, salinity float64 */
#define MLX_DTYPE_T64_S32 4 /* theta float64, salinity float32 */

/* flags of the fused steric entry points */
#define MLX_FLAG_SKIP_DRY 1 /* do not load theta/S where the reference volcello (K1) / rho0m (K2)
                               is NaN: those cells contribute exactly nothing (a NaN product is
                               skipped) resp. are NaN whatever theta/S hold, so the results are
                               bit-identical while whole cache lines of land and sub-bottom cells
                               never leave HBM.  0 = load every cell, wet or dry.               */
#define MLX_FLAG_FMA 2      /* fused arithmetic for the Wright DENSITY: the reference's expression
                               tree with every "c + a*b" contracted into one fma and the quotient
                               taken from the hardware reciprocal seed and a cubic correction
                               (<= 1 ulp).  NOT bit-identical to numpy:
                               - MLX_DTYPE_F64: rho differs by a few ulp (parity gate: 1e-10 relative
                                 on rho and masso, 1e-10*max|ref| on delta_rho and eta);
                               - MLX_DTYPE_F32 (numpy's mixed precision): the float32 polynomial is
                                 evaluated exactly as numpy rounds it, only the float64 tail
                                 (lam + al0*(p+p0), the quotient) is fused: a few float64 ulp from what
                                 numpy computes on float32 input;
                               - MLX_DTYPE_F32_UPCAST: the float64 form on the upcast values (a few
                                 ulp from float64 arithmetic on them, ~1e-7 from numpy's float32
                                 polynomial).
                               All kernels share one tree, so in this mode too masso(t=0) == masso0
                               and delta_rho(t=0) == 0 hold exactly.  About two thirds of the VALU work
                               per cell: lifts the thermosteric / halosteric sums and every float32
                               sum off the fp64-VALU bound.  momlevel_amd passes it by default for the
                               global sums (mlx_steric_global*) only.  Denominators of exactly 0 / inf
                               / denormal size (never sea water) give NaN here, inf / 0 in numpy.  */
#define MLX_FLAG_TCHUNK_MASK 0xFF00 /* K1 tuning hint, never changes a result: time steps per
                               block = 8 * ((flags >> 8) & 0xFF); 0 = the default (32)            */
#define MLX_FLAG_TCHUNK(steps) ((((steps) / 8) & 0xFF) << 8)

int mlx_version(void);
/* copies the calling thread's last error text into buf (NUL-terminated); returns its length */
int mlx_last_error(char *buf, size_t n);

/* which build of this ABI the library is: the product binds MLX_BUILD_HIP only (momlevel_amd/_lib.py
 * refuses anything else); MLX_BUILD_HOST is the CPU restatement used as a checker by the tests
 * (oracle/host_abi.c), which must never stand in for the device library. */
#define MLX_BUILD_HIP  1
#define MLX_BUILD_HOST 2
int mlx_build_kind(void);

/* Name of the kernel instantiation the calling thread's last mlx_steric_global* / mlx_steric_local*
 * call launched, with its template arguments, e.g. "k_steric_global<double,2,4,0,0,false,false,true>"
 * = <element type, cells per 16-byte pack, packs (K1) or time steps (K2) per thread, variant
 * (0 steric, 1 halosteric, 2 thermosteric, 3 all), dtype mode, generic twin, MLX_FLAG_SKIP_DRY,
 * MLX_FLAG_FMA> -- what a profile of that call lists.  bench.py quotes it; same contract as
 * mlx_last_error (copies into buf, returns the length). */
int mlx_last_kernel(char *buf, size_t n);

/* ---------------------------------------------------------------------------------
 * K0  pointwise EOS map.  Replaces eos.wright.density/drho_dtemp/drho_dsal/alpha/beta
 * (src/momlevel/eos/wright.py:23-165), eos.linear.density/alpha/beta and its constant
 * derivatives (eos/linear.py:26-162) and
 * the apply_ufunc in derived.calc_rho (src/momlevel/derived.py:621-630).
 * out[t,z,i] = f(T[t*t_stride_T + z*plane + i], S[t*t_stride_S + z*plane + i], p).
 * t_stride_* == 0 broadcasts a (z,y,x) field over time (the held field of the
 * thermosteric / halosteric variants).  out is always float64.
 * ------------------------------------------------------------------------------- */
int mlx_eos_map(const void *T, const void *S, int dtype,
                const double *p, int p_mode, int eos, int func,
                int64_t nt, int64_t nz, int64_t plane,
                int64_t t_stride_T, int64_t t_stride_S, int flags, /* 0 or MLX_FLAG_FMA (density) */
                double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * K0 for EVERY dtype combination numpy accepts.  The reference's EOS functions are bare numpy
 * expressions (src/momlevel/eos/wright.py:44-48, 74-83, 108-117, 142, 165; eos/linear.py:44-46,
 * 104, 124), so their sub-expressions take the dtypes numpy's promotion gives them (numpy >= 2,
 * NEP 50): array (op) array -> the wider dtype; a python float takes the dtype of the array it
 * meets.  mlx_eos_map covers float64 fields and float32 theta/S with a float64 pressure; this
 * entry point covers the rest bit for bit -- a python-float or float32 pressure on float32 fields
 * (the whole expression stays float32: derived.calc_pdens, src/momlevel/derived.py:477, on MOM6's
 * float32 output), theta and salinity of different dtypes, python floats for theta or salinity.
 *
 * Each operand is n cells (stride 1) or one value used for every cell (stride 0) of kind
 *   MLX_KIND_F64 / MLX_KIND_F32: a DEVICE pointer to float64 / float32 (numpy scalars count as
 *                                arrays of their dtype);
 *   MLX_KIND_WEAK:               a python float or int -- a HOST pointer to ONE double, read
 *                                during the call; stride ignored.
 * func: any MLX_FUNC_* (MLX_FUNC_IBH: out = p * (-1.0 / (rho * gravity)), dynamic.py:34-36, gravity a
 * python float; ignored otherwise).  The linear EOS never reads p: it may be NULL there and takes no
 * part in the promotion -- except for MLX_FUNC_DENSITY_REF (MLX_EOS_LINEAR only, MLX_E_ENUM
 * otherwise): eos.linear.density(T, S, p, rho_ref) with rho_ref given, whose constant term
 * c = RHO_T0_S0 - rho_ref (src/momlevel/eos/linear.py:55; a python float, or a numpy scalar when
 * rho_ref is one) arrives IN THE p OPERAND with its kind, and out = c + ((-0.2*T) + (0.8*S)) under
 * numpy's promotion (:56).  The result comes back IN NUMPY'S RESULT DTYPE: out is a device buffer of
 * n*8 bytes, 8-byte aligned; *out_kind (host, required) receives MLX_KIND_F32 when numpy's result
 * is float32 -- out then holds n float32 values (its first n*4 bytes) -- and MLX_KIND_F64 when out
 * holds n float64 values.  (The kind depends on the operand kinds, eos and func only.)
 * ------------------------------------------------------------------------------- */
#define MLX_KIND_F64  0
#define MLX_KIND_F32  1
#define MLX_KIND_WEAK 2
int mlx_eos_map_promote(const void *T, int kind_T, int64_t stride_T,
                        const void *S, int kind_S, int64_t stride_S,
                        const void *p, int kind_p, int64_t stride_p,
                        int eos, int func, double gravity, int64_t n,
                        void *out, int *out_kind, void *stream);

/* ---------------------------------------------------------------------------------
 * dynamic.inverse_barometer (src/momlevel/dynamic.py:34-36): out = p * (-1.0 / (rho(T,S,p) * gravity))
 * on the same grid conventions as mlx_eos_map (the reference calls it on (time,yh,xh) surface
 * fields: pass nz = 1).  p is both the EOS pressure and the numerator.
 * ------------------------------------------------------------------------------- */
int mlx_inverse_barometer(const void *T, const void *S, int dtype,
                          const double *p, int p_mode, int eos, double gravity,
                          int64_t nt, int64_t nz, int64_t plane,
                          int64_t t_stride_T, int64_t t_stride_S,
                          double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * K1  fused EOS + rho*vol0 + sum over (z,y,x) per time step: masso_out[t].
 * Replaces calc_rho followed by calc_masso (src/momlevel/derived.py:597-639, 414-444)
 * as called from steric() for domain="global" (src/momlevel/steric.py:128,135) and
 * from setup_reference_state (src/momlevel/reference.py:71,77).
 * vol0 is the REFERENCE volcello (z,y,x), used for every time step (steric.py:135).
 * p_mode: any (MLX_P_FULL4D = a time-dependent patm, steric.py:58-60,96).
 * t_stride_T == 0 (or t_stride_S == 0) holds that field at its (z,y,x) reference slab for every
 * time step: the halosteric (thermosteric) variant, steric.py:115-125.
 * In a multi-GPU run each rank passes its horizontal tile; the caller all-reduces
 * masso_out (RCCL) -- the library never communicates.
 * ------------------------------------------------------------------------------- */
size_t mlx_steric_global_workspace_bytes(int64_t nt, int64_t nz, int64_t plane);
int mlx_steric_global(const void *T, const void *S, int dtype,
                      const double *vol0, const double *p, int p_mode, int eos,
                      int64_t nt, int64_t nz, int64_t plane,
                      int64_t t_stride_T, int64_t t_stride_S, int flags,
                      double *masso_out, void *workspace, size_t workspace_bytes,
                      void *stream);

/* ---------------------------------------------------------------------------------
 * K1, all variants in ONE pass over theta/S (BASELINE.json configs[4]): what three calls of
 * steric(..., domain="global") with variant = "steric", "thermosteric", "halosteric"
 * (src/momlevel/steric.py:115-147, 187-196) compute, from a single read of the 4-D fields:
 *   out[0*nt + t] = sum rho(T[t], S[t], p) * vol0      (steric)
 *   out[1*nt + t] = sum rho(T[t], S0,   p) * vol0      (thermosteric: S held at the reference)
 *   out[2*nt + t] = sum rho(T0,   S[t], p) * vol0      (halosteric:   theta held)
 *   out[3*nt + t] = sum T[t] * vol0                    (EXTENSION, not in momlevel: the ocean heat
 *                   content integrand; OHC(t) = rho0 * c_p * out[3*nt+t], scaled by the caller)
 * T0, S0: the (z,y,x) reference slabs (same dtype as T, S).  Each of the first three rows is
 * bit-identical to the corresponding mlx_steric_global call (same tiling, same order of
 * summation).  out holds 4*nt doubles; workspace: mlx_steric_global_decomp_workspace_bytes().
 * ------------------------------------------------------------------------------- */
size_t mlx_steric_global_decomp_workspace_bytes(int64_t nt, int64_t nz, int64_t plane);
int mlx_steric_global_decomp(const void *T, const void *S, const void *T0, const void *S0,
                             int dtype, const double *vol0, const double *p, int p_mode, int eos,
                             int64_t nt, int64_t nz, int64_t plane,
                             int64_t t_stride_T, int64_t t_stride_S, int flags,
                             double *out, void *workspace, size_t workspace_bytes,
                             void *stream);

/* ---------------------------------------------------------------------------------
 * K2  fused EOS + delta_rho + dz-weighted column integral.
 * Replaces steric.py:151-166 for domain="local":
 *   delta_rho = where(vol0 notnull, rho - rho0, NaN)             (nt,nz,plane)
 *   eta       = (-1/rhozero) * sum_z(dz * delta_rho) [skipna],   (nt,plane)
 *               NaN where vol0[z=0] is NaN
 * rho0m is rho0 with vol0's NaN mask folded in -- make it once per reference
 * state with mlx_fold_mask().  vol0_surface is the z=0 plane of the reference
 * volcello.  dz: either an explicit (nz,plane) array, or NULL -- then the kernel
 * evaluates calc_dz(levels, z_i, deptho) with its default top/bottom
 * (src/momlevel/derived.py:295-318) from z_i[nz+1] and deptho[plane] on the fly.
 * delta_rho_out may be NULL: the 8 B/cell store is then skipped.
 * neg_inv_rhozero is (-1.0/rhozero) evaluated by the caller in float64.
 * ------------------------------------------------------------------------------- */
int mlx_fold_mask(const double *rho0, const double *vol0, int64_t n,
                  double *rho0m_out, void *stream);
int mlx_steric_local(const void *T, const void *S, int dtype,
                     const double *rho0m, const double *vol0_surface,
                     const double *dz, const double *z_i, const double *deptho,
                     const double *p, int p_mode, int eos, double neg_inv_rhozero,
                     int64_t nt, int64_t nz, int64_t plane,
                     int64_t t_stride_T, int64_t t_stride_S, int flags,
                     double *delta_rho_out, double *eta_out, void *stream);

/* ---------------------------------------------------------------------------------
 * K2, all variants in ONE pass over theta/S: what three calls of steric(..., domain="local")
 * with variant = "steric", "thermosteric", "halosteric" compute (src/momlevel/steric.py:115-125,
 * 150-166), from a single read of the 4-D fields (16 B read + 3 x 8 B written per cell instead of
 * 3 x (16|8 read + 8 written)).  T0, S0: the (z,y,x) reference slabs (same dtype as T, S).
 * Variant v (0 steric, 1 thermosteric, 2 halosteric) lands at
 *   delta_rho_out + v*delta_rho_variant_stride   (nt,nz,plane)   [delta_rho_out may be NULL]
 *   eta_out       + v*eta_variant_stride         (nt,plane)
 * (strides in elements, >= the size of one field).  Each field is bit-identical to the
 * corresponding mlx_steric_local call.
 * ------------------------------------------------------------------------------- */
int mlx_steric_local_decomp(const void *T, const void *S, const void *T0, const void *S0, int dtype,
                            const double *rho0m, const double *vol0_surface,
                            const double *dz, const double *z_i, const double *deptho,
                            const double *p, int p_mode, int eos, double neg_inv_rhozero,
                            int64_t nt, int64_t nz, int64_t plane,
                            int64_t t_stride_T, int64_t t_stride_S, int flags,
                            double *delta_rho_out, int64_t delta_rho_variant_stride,
                            double *eta_out, int64_t eta_variant_stride, void *stream);

/* ---------------------------------------------------------------------------------
 * skipna sum of n float64 values -> out[0].  Replaces volcello.sum() in
 * derived.calc_volo (src/momlevel/derived.py:789) and areacello.sum() in
 * util.validate_areacello (src/momlevel/util.py:692) / steric.py:138.
 * ------------------------------------------------------------------------------- */
size_t mlx_nansum_workspace_bytes(int64_t n);
int mlx_nansum(const double *x, int64_t n, double *out,
               void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------
 * sum(rho*vol) per time step for an ALREADY MATERIALISED rho: the standalone
 * derived.calc_masso (src/momlevel/derived.py:414-444).  vol_t_stride == 0
 * broadcasts a (z,y,x) volcello over time; otherwise volcello is 4-D like rho.
 * n3 = nz*plane.  Workspace: mlx_steric_global_workspace_bytes(nt, 1, n3).
 * ------------------------------------------------------------------------------- */
int mlx_masso(const double *rho, const double *vol, int64_t nt, int64_t n3,
              int64_t vol_t_stride, double *masso_out,
              void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------
 * util.annual_average (src/momlevel/util.py:85-92): per-group weighted mean over the leading
 * (time) axis, groups of group_len consecutive steps (12 monthly steps per year), weights =
 * days in month.  x is (ngroups*group_len, n), w (ngroups*group_len), out (ngroups, n):
 *   out[g,i] = sum_j nan0(x[gL+j,i]) * w[gL+j]  /  sum_j [x[gL+j,i] not NaN] * w[gL+j]
 * (xarray's weighted mean: NaNs carry no weight; 0 total weight gives NaN).  Terms are added
 * in time order.
 * ------------------------------------------------------------------------------- */
int mlx_group_weighted_mean(const double *x, const double *w, int64_t ngroups, int64_t group_len,
                            int64_t n, double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * derived.calc_dz (src/momlevel/derived.py:249-325): dz_out[z,i] from z_i[nz+1] and
 * depth[plane] (NaN depth -> 0).  has_bottom=0 ignores `bottom`.  fraction != 0
 * returns the cell fraction (NaN where dz or the cell thickness is 0).
 * The sign checks (derived.py:284-292) stay on the host.
 * ------------------------------------------------------------------------------- */
int mlx_calc_dz(const double *z_i, const double *depth, int64_t nz, int64_t plane,
                double top, double bottom, int has_bottom, int fraction,
                double *dz_out, void *stream);

/* ---------------------------------------------------------------------------------
 * Measurement aid (not a reference function): out[i] = a[i] + b[i] over n doubles (n even, all
 * pointers 16-byte aligned) with the 16-byte non-temporal loads and stores of the fused local
 * kernel -- 16 B read + 8 B written per element and no arithmetic to speak of.  bench.py reports
 * mlx_steric_local with delta_rho (24 B/cell) as a fraction of this box-specific ceiling.
 * ------------------------------------------------------------------------------- */
int mlx_stream_probe(const double *a, const double *b, int64_t n, double *out, void *stream);

/* The same aid for the read:write mixes of the other local passes: one (b == NULL) or two streams
 * of n elements of `dtype` (MLX_DTYPE_F64 or MLX_DTYPE_F32; 16-byte aligned, n a whole number of
 * 16-byte packs) in and, if write_out != 0, out[i] = (double)a[i] (+ (double)b[i]) as one float64
 * stream out (16-byte aligned, n doubles).  write_out == 0: read-only -- out must hold one double
 * and is left untouched.  A held-field pass with delta_rho is 1 in / 1 out (8 + 8 B per cell at
 * float64, 4 + 8 B at float32), the float32 steric pass with delta_rho 2 in / 1 out (8 + 8 B).
 * bench.py reports each local instantiation as a fraction of the probe with ITS mix. */
int mlx_stream_probe_mix(const void *a, const void *b, int dtype, int64_t n, double *out,
                         int write_out, void *stream);

/* Measurement aid: the box's float64 vector-ALU issue rate.  Launches a kernel of nothing but
 * independent v_fma_f64 chains (8192 blocks x 256 threads x 8 chains x iters) and stores the number
 * of lane-instructions it issues in *lane_instructions (host memory); the caller times it.
 * bench.py prices the instruction-issue-bound kernels (float32 inputs, exact held-field sums, the
 * one-pass kernels) against this ceiling.  out: one double of device memory, left untouched. */
int mlx_valu_probe(int64_t iters, double *out, int64_t *lane_instructions, void *stream);

/* ---------------------------------------------------------------------------------
 * Synthetic MOM6-shaped fields for bench.py and the full-size tests (not a
 * reference function).  out[t,z,y,x] = lo + scale*u, with
 * u = (splitmix64(seed ^ (field_id << 60) ^ gidx) >> 11) * 2^-53 and gidx the
 * linear index of the cell in the GLOBAL (NT?,nz,NY,NX) grid, so that a
 * horizontal tile (y0,x0,ny,nx) of a rank reproduces its part of the global field.
 * Cells where mask3d[z,y,x] is NaN get NaN (mask3d may be NULL).
 * ------------------------------------------------------------------------------- */
int mlx_synth_field(void *out, int dtype, int64_t nt, int64_t nz, int64_t ny, int64_t nx,
                    int64_t t0, int64_t NY, int64_t NX, int64_t y0, int64_t x0,
                    uint64_t seed, int field_id, double lo, double scale,
                    const double *mask3d, void *stream);

/* ---------------------------------------------------------------------------------
 * Stratification diagnostics: the consumers of alpha / beta.
 *
 * mlx_stratification -- derived.calc_n2 (derived.py:328-411, interfaces=None) and
 * derived.calc_stability_angle (derived.py:714-766) in one pass over theta / S:
 *   MLX_STRAT_N2      gravity * ((alpha * dT/dz) - (beta * dS/dz))
 *   MLX_STRAT_TURNER  degrees(arctan((1 + R) / (1 - R))),  R = (beta * dS/dz) / (alpha * dT/dz)
 * with alpha, beta of the chosen EOS at (T, S, p) and d/dz = numpy.gradient(f, z, edge_order=2)
 * along z (xarray's differentiate).  T, S, out: (nt, nz, plane); dtype MLX_DTYPE_F64, _F32
 * (numpy's mixed precision: the derivative of a float32 field is float32, alpha / beta float64;
 * Wright only) or _F32_UPCAST.  p[t*p_stride_t + z*p_stride_z + cell*p_stride_cell] in elements
 * (a z profile: 0, 1, 0; a scalar: 0, 0, 0; a full field: nz*plane, plane, 1); NULL for the linear
 * EOS.  coef: (nz, 3) device doubles, numpy.gradient's a, b, c of every level computed by the
 * caller from the coordinate exactly as numpy does (rows 0 and nz-1: the one-sided formulas on
 * levels (0,1,2) / (nz-3,nz-2,nz-1)); uniform != 0 (numpy: all spacings equal): interior levels are
 * (f[k+1] - f[k-1]) / two_dx instead.  nz >= 3.  Values identical to numpy's except the arctan.
 *
 * mlx_adjust_negative_n2 -- derived.adjust_negative_n2 (derived.py:30-71) and the column sum of
 * derived.calc_wave_speed (derived.py:798-831) in one walk down each column.  n2, adjusted:
 * (nt, nz, plane), nt = the product of the dimensions before z.  The reference's
 * `adjusted[0].fillna(1e-8)` indexes the array's LEADING dimension: lead0_rows > 0 = that is not z,
 * and the first lead0_rows of the nt rows (index 0 of the leading dimension: 1 for a
 * (time, z, y, x) field) are filled at every level; lead0_rows == 0 (nt == 1) = z leads and the
 * surface level is filled.  speed (nt, plane) = sum_z sqrt(adjusted) * dz / pi (skipna), dz
 * (nz, plane); with lead0_rows == 0 also NaN where the surface n2 is NaN.  adjusted or speed may
 * be NULL.
 *
 * mlx_wave_speed_where_time0 -- what derived.py:823 returns for a (time, z, y, x) n2: xarray
 * broadcasts the condition n2[time=0] (z, y, x) against the speeds (time, y, x) into
 * out (nz, plane, nt).
 * ------------------------------------------------------------------------------- */
#define MLX_STRAT_N2     0
#define MLX_STRAT_TURNER 1
int mlx_stratification(const void *T, const void *S, int dtype, const double *p,
                       int64_t p_stride_t, int64_t p_stride_z, int64_t p_stride_cell,
                       int eos, int func, const double *coef, int uniform, double two_dx,
                       double gravity, int64_t nt, int64_t nz, int64_t plane, double *out,
                       void *stream);
int mlx_adjust_negative_n2(const double *n2, int64_t nt, int64_t nz, int64_t plane,
                           int64_t lead0_rows, const double *dz, double *adjusted, double *speed,
                           void *stream);
int mlx_wave_speed_where_time0(const double *n2_t0, const double *speed, int64_t nt, int64_t nz,
                               int64_t plane, double *out, void *stream);

/* ---------------------------------------------------------------------------------
 * Host side of the staged transfers (momlevel_amd/hostio.py; no reference counterpart: the
 * reference passes numpy arrays to numpy).  Copies nbytes of HOST memory from src to dst, split
 * into page-aligned slices over `threads` (1..64) threads: the caller's and threads-1 workers of a
 * team the library keeps (created on first use, shared by concurrent callers, re-created in a
 * forked child).  Returns when dst is complete.  streaming != 0 writes dst with non-temporal stores
 * (no read-for-ownership of the destination lines -- for data that is touched once: a result
 * array, a staging buffer the DMA engine reads next).  The ranges must not overlap (MLX_E_SHAPE).
 * Touches no device; callable concurrently on disjoint ranges.  One call per staging piece from
 * Python: a foreign call, so the GIL is released for its whole duration.
 * ------------------------------------------------------------------------------- */
int mlx_host_copy(void *dst, const void *src, size_t nbytes, int threads, int streaming);

/* dst[i] = mask[i] ? NaN : src[i] for n elements of elem_size 4 (float32) or 8 (float64) bytes of
 * HOST memory -- bit patterns are copied, the canonical quiet NaN is written where the mask byte is
 * non-zero -- split over the same team of `threads` (1..64) threads.  How a numpy masked array (what
 * netCDF4.Variable.__getitem__ returns where a file declares _FillValue) becomes the NaN-filled
 * array the reference is handed by xarray (src/momlevel/steric.py:84-96 sees DataArrays only;
 * momlevel_amd/labeled.py as_plain), at host memory bandwidth.  dst / src element-aligned
 * (MLX_E_ALIGN); dst must overlap neither src nor mask (MLX_E_SHAPE); elem_size other than 4 / 8:
 * MLX_E_ENUM.  Touches no device. */
int mlx_host_copy_masked(void *dst, const void *src, const unsigned char *mask, size_t n,
                         int elem_size, int threads);

#ifdef __cplusplus
}
#endif
#endif /* MOMLEVEL_HIP_H */
