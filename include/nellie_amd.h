/*
 * nellie_amd.h -- C-ABI of libnellie_hip.so, the MI355X (gfx950) native engine for
 * Nellie's `nellie/segmentation` hot path: Filter (multiscale Frangi) -> Label.
 *
 * The reference (aelefebv/nellie v1.0.3) is pure Python: its "FFI" for this path is the
 * `xp`/`ndi` module handle that swaps numpy/scipy.ndimage for cupy/cupyx
 * (nellie/segmentation/filtering.py:117-159, labelling.py:115-154).  This library is what
 * a third backend binds instead: every entry point below replaces one group of xp/ndi calls
 * of the reference, cited as `file:line` relative to the reference checkout.  The host side
 * stays Python (the nellie_amd.segmentation modules) and talks to this ABI through ctypes:
 * numpy in, numpy out, plain pointers and sizes, no torch / cupy types.
 *
 * Conventions
 *  - Every function returns an int status (NL_OK == 0) and, on failure, writes a
 *    NUL-terminated message into the caller's `err` buffer (`errlen` bytes, may be NULL/0).
 *    No exceptions cross the boundary.  The Python layer maps NL_ENODEV to
 *    RuntimeError("GPU backend requested but ...") and NL_ENOMEM to MemoryError so the
 *    reference's retry ladder (nellie/utils/adaptive_run.py:116-141) keeps working.
 *  - Volumes are C-contiguous (Z, Y, X), X fastest.  All host buffers are owned by the
 *    caller; all device buffers are owned by the context.  Inputs are never written.
 *  - A context is bound to one device and one local volume shape.  It may hold a Z-slab of
 *    a larger global volume (multi-GPU): `gz0` is the global index of local plane 0, `gnz`
 *    the global plane count, `[own_lo, own_hi)` the planes this rank owns (the rest are
 *    ghost planes).  Boundary rules (reflect padding, one-sided differences, zero border)
 *    apply only at true faces of the GLOBAL volume.  Single GPU: gz0 = 0, gnz = nz,
 *    own = [0, nz).
 *  - Entry points call hipSetDevice themselves and are synchronous on return unless noted
 *    (nl_* _async variants enqueue on the context stream; nl_sync waits).  One context must
 *    not be used from two threads at once; different contexts may be.  The one exception are
 *    the copy-thread calls of the frame streamer -- nl_input_load_async, nl_input_wait, nl_outputs_fetch_async,
 *    nl_outputs_wait -- which touch only the copy streams, the input slots and the staging
 *    volumes and may run beside the compute thread's calls on the same context.
 */
#ifndef NELLIE_AMD_H
#define NELLIE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes */
#define NL_OK      0
#define NL_EINVAL  1   /* bad argument */
#define NL_ENODEV  2   /* no usable HIP device */
#define NL_ENOMEM  3   /* device or host allocation failed (out of memory) */
#define NL_EHIP    4   /* any other HIP runtime error */
#define NL_ESTATE  5   /* call order violated */
#define NL_ECOMM   6   /* RCCL error */

/* numpy dtype codes accepted by the upload entry points */
#define NL_U8 0
#define NL_I8 1
#define NL_U16 2
#define NL_I16 3
#define NL_U32 4
#define NL_I32 5
#define NL_F32 6
#define NL_F64 7
#define NL_U64 8
#define NL_I64 9

/* fields a sampling call can read */
#define NL_FIELD_GAUSS  0  /* current Gaussian scale-space volume */
#define NL_FIELD_FROB   1  /* sqrt(frob_sq)/max_abs of the current scale, inf -> max finite (filtering.py:421-426, 562) */
#define NL_FIELD_FRANGI 2  /* the Filter output (after nl_filter_finish / nl_mask_volume) or the uploaded Frangi image */
#define NL_FIELD_VESSELNESS 3 /* vesselness * masks BEFORE nl_filter_finish: the running maximum seen through the
                                 cumulative mask bits (what NL_FIELD_FRANGI would hold after nl_filter_finish) */

typedef struct nl_ctx nl_ctx;

const char *nl_version(void);

/* Replaces cupy.cuda.runtime.getDeviceCount / memGetInfo (adaptive_run.py:23-43). */
int nl_device_count(int *count, char *err, size_t errlen);
int nl_device_mem_info(int device, int64_t *free_bytes, int64_t *total_bytes, char *err, size_t errlen);
int nl_device_name(int device, char *name, size_t namelen, char *err, size_t errlen);

int nl_ctx_create(nl_ctx **out, int device,
                  int64_t nz_local, int64_t ny, int64_t nx,
                  int64_t gz0, int64_t gnz, int64_t own_lo, int64_t own_hi,
                  char *err, size_t errlen);
int nl_ctx_destroy(nl_ctx *ctx);
int nl_sync(nl_ctx *ctx, char *err, size_t errlen);
/* bytes of device memory a context of this shape allocates (for the low-memory decision,
   adaptive_run.py:88-100) */
int64_t nl_ctx_bytes(int64_t nz_local, int64_t ny, int64_t nx);

/* ------------------------------------------------------------------ Filter ------------ */

/* frame = xp.asarray(memmap[t], dtype=float32)  (filtering.py:917-924).
   Uploads local planes [z0, z1) from `host` (shape (z1-z0, ny, nx), dtype code) and
   converts to float32.  Also resets the per-frame state (vesselness = 0, masks = 1;
   filtering.py:807-808). */
int nl_filter_load(nl_ctx *ctx, const void *host, int dtype, int64_t z0, int64_t z1,
                   char *err, size_t errlen);

/* The same in two halves, so a caller can keep the raw frame resident in HBM and restart from
   it without touching the host again (what the reference's cupy path gets from keeping `frame`
   alive): nl_input_load = H2D of the raw planes (any dtype, extra device allocation of the
   frame's own size); nl_filter_begin = float32 conversion + per-frame reset, asynchronous. */
int nl_input_load(nl_ctx *ctx, const void *host, int dtype, int64_t z0, int64_t z1,
                  char *err, size_t errlen);
int nl_filter_begin(nl_ctx *ctx, char *err, size_t errlen);

/* One cascade step: ndi.gaussian_filter(gauss, sigma=delta, output=gauss, mode="reflect",
   truncate=3.0) (filtering.py:827-835) = up to three correlate1d passes, axis order Z, Y, X,
   float64 accumulation in scipy's symmetric order, float32 store after each pass.
   w* are the 2r+1 float64 kernel weights the host computed exactly like scipy
   (`_gaussian_kernel1d`); a NULL pointer skips that axis (sigma <= 1e-15).
   Output planes [z0, z1) are produced; the input must be valid on [z0-rz, z1+rz) clipped to
   the local slab (true global faces reflect). */
int nl_gauss_step(nl_ctx *ctx,
                  const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                  int64_t z0, int64_t z1, char *err, size_t errlen);

/* The same step enqueued AHEAD on the context's side stream: the cascade step of scale s+1 only reads the Gaussian of
   scale s, so it can overlap the Hessian walk of scale s.  The result becomes NL_FIELD_GAUSS at nl_gauss_commit
   (call it before anything of scale s+1).  In between, only the per-scale calls of Filter are allowed (sampling by
   min/max + histogram, nl_hessian_stats, nl_vesselness_*); nl_sample_gather, nl_mask_volume* and Label use the free
   Gaussian volumes as scratch.
   Only steps that write at most TWO of the three ping-pong volumes can run ahead (a Z pass + a fused Y+X pass: equal
   in-plane radii up to nl_ctx_info("gauss_yx_max_r")); a step of one pass per axis would write its third pass into the
   volume the current scale reads and is refused with NL_ESTATE -- run it with nl_gauss_step, in order. */
int nl_gauss_step_ahead(nl_ctx *ctx, const double *wz, int rz, const double *wy, int ry, const double *wx, int rx,
                        int64_t z0, int64_t z1, char *err, size_t errlen);
int nl_gauss_commit(nl_ctx *ctx, char *err, size_t errlen);

/* arr[::sz, ::sy, ::sx] of a field on the owned planes (strides are those of the GLOBAL
   lattice; filtering.py:342-346, 355-356).  `out` receives the samples in C order, `*n`
   their number; `cap` is the capacity of `out` in elements. */
int nl_sample_gather(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx,
                     float *out, int64_t cap, int64_t *n, char *err, size_t errlen);

/* nl_sample_gather restricted to the positive samples, compacted on the device so that only they cross PCIe (every
   consumer takes arr[arr > 0] first: filtering.py:357, 957-959).  Order unspecified; `cap` >= number of lattice
   points; *n = number of positives written. */
int nl_sample_gather_positive(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx,
                              float *out, int64_t cap, int64_t *n, char *err, size_t errlen);
/* The same in two halves: _begin enqueues the compaction (and reports the number of lattice points = the capacity _end
   needs at most), _end waits and fetches.  Between the two the host is free; the only calls allowed on this context in
   between are nl_chain_finish / nl_chain_log (which enqueue nothing). */
int nl_sample_gather_positive_begin(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx, int64_t *n_lattice, char *err, size_t errlen);
int nl_sample_gather_positive_end(nl_ctx *ctx, float *out, int64_t cap, int64_t *n, char *err, size_t errlen);

/* min / max / count of the POSITIVE lattice samples (arr[arr > 0]; filtering.py:357 and the
   range=(min,max) of gpu_functions.py:31-35, 60-64).  npos == 0 leaves mn/mx untouched. */
int nl_sample_minmax(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx,
                     float *mn, float *mx, int64_t *npos, char *err, size_t errlen);

/* numpy.histogram(positive, bins=nbins, range=(first,last)) counts (gpu_functions.py:31, 60):
   float32 index computation + the +-1 correction against the float32 `edges` (nbins+1
   values from numpy.linspace, computed by the host).  counts[nbins] int64. */
int nl_sample_hist(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx,
                   const float *edges, int nbins, int64_t *counts, char *err, size_t errlen);

/* nl_sample_minmax followed by nl_sample_hist without a host round trip in between: numpy's float32 bin edges for
   range = (min, max) of the positive samples (`_get_outer_edges` / `np.linspace`, numpy/lib/_histograms_impl.py) are
   formed on the device.  *valid: 0 = no positive sample (counts are zero), 1 = ok, 2 = the range is not finite (numpy
   raises ValueError there; counts are zero).  `edges` (may be NULL) receives the nbins + 1 edges the device used;
   they equal np.linspace(np.float32(min), np.float32(max), nbins + 1, dtype=float32) bit for bit. */
int nl_sample_range_hist(nl_ctx *ctx, int field, int64_t sz, int64_t sy, int64_t sx, int nbins, float *mn, float *mx,
                         int64_t *n_positive, int64_t *counts, float *edges, int *valid, char *err, size_t errlen);
/* The same for two fields that need nothing from each other (the gamma samples of the Gaussian and the raw Frobenius
   samples of a scale: filtering.py:365-380, 421-444) in ONE device round trip; every output is an array of two (counts:
   2 x nbins, edges: 2 x (nbins + 1)), [0] = field_a. */
int nl_sample_range_hist2(nl_ctx *ctx, int field_a, int field_b, int64_t sz, int64_t sy, int64_t sx, int nbins, float *mn, float *mx,
                          int64_t *npos, int64_t *counts, float *edges, int *valid, char *err, size_t errlen);

/* The two histogram thresholds of nellie/utils/gpu_functions.py on a finished histogram (host arithmetic, no device
   work; callable without a context): `counts[nbins]` int64 as numpy.histogram returns them, `edges[nbins + 1]`
   float32 bin edges.  *otsu = centre of the bin maximising the between-class variance (gpu_functions.py:36-50),
   *triangle = the triangle-method bin centre (:64-94), both float32 values widened to double, computed with numpy's
   float64 operation order so that they carry numpy's bits.  *status: 0 ok, 1 = the triangle construction is degenerate
   (numpy raises "attempt to get argmax of an empty sequence" there; *triangle is 0, *otsu is valid). */
int nl_hist_thresholds(const int64_t *counts, const float *edges, int nbins, double *triangle, double *otsu, int *status,
                       char *err, size_t errlen);
/* The same for either edge type numpy.histogram produces: edges_f64 = 0 -> `edges` is float32[nbins + 1] (float32 data), 1 ->
   float64[nbins + 1] (integer and float64 data -- Label's intensity thresholds on the original image, labelling.py:414-431:
   the bin centres are then float64 like the reference's).  *otsu_var (may be NULL) = the between-class variance at the
   threshold, the second value of gpu_functions.py:50. */
int nl_hist_thresholds_ex(const int64_t *counts, const void *edges, int edges_f64, int nbins, double *triangle, double *otsu,
                          double *otsu_var, int *status, char *err, size_t errlen);
/* Host only: np.histogram(values, bins=nbins) of float32 data (range = its min .. max) and nl_hist_thresholds_ex of it in one
   call -- Label's log-domain threshold (labelling.py:448-455) on the few 10^4 samples the device compacted.  Same float32
   arithmetic as the device histogram.  *status: 0 ok, 1 degenerate triangle, 2 range not finite (numpy raises ValueError for
   both).  counts_out (nbins) / edges_out (nbins + 1): optional copies of the histogram. */
int nl_host_hist_thresholds_f32(const float *values, int64_t n, int nbins, double *triangle, double *otsu, int *status,
                                int64_t *counts_out, float *edges_out, char *err, size_t errlen);

/* Hessian by double finite differences of the current Gaussian volume (xp.gradient twice,
   filtering.py:518-536) on the owned planes; returns
     max_abs          = max over the six components of max|h|         (filtering.py:556-561; NOT yet mapped 0 -> 1)
     max_frob_sq      = largest finite frob_sq                         (filtering.py:538-543)
     any_inf          = 1 if some frob_sq is +inf                      (filtering.py:421-426)
   `spacing` = (dz, dy, dx) as the reference's python floats. */
int nl_hessian_stats(nl_ctx *ctx, const double spacing[3],
                     float *max_abs, float *max_frob_sq, int *any_inf, char *err, size_t errlen);

/* Fixes the normalisation NL_FIELD_FROB and nl_vesselness_step use: frob = sqrt(frob_sq)/max_abs,
   +inf replaced by max_finite (the host passes max_abs already mapped <=0 -> 1.0). */
int nl_set_frob_norm(nl_ctx *ctx, float max_abs, float max_finite, char *err, size_t errlen);

/* One scale of filtering.py:842-851 on the owned planes:
     h_mask     = use_thr ? frob > thr : frob > 0            (filtering.py:428-444)
     v          = Frangi(eigvalsh(H) sorted by |.|) on h_mask, 0 elsewhere (filtering.py:574-585, 744-766)
     vesselness = maximum(vesselness, v); masks &= h_mask    (filtering.py:850-851)
   The host calls it only when the mask is non-empty (filtering.py:843-844).
   Planes [z0, z1) are processed (z0 = z1 = -1: the owned planes; a slab passes own +-2 so that the
   opening of nl_mask_volume finds valid neighbours).  `mask_count` (may be NULL) receives the number
   of OWNED voxels in h_mask. */
int nl_vesselness_step(nl_ctx *ctx, float gamma_sq, float alpha_sq, float beta_sq,
                       int use_thr, float thr, int64_t z0, int64_t z1, int64_t *mask_count,
                       char *err, size_t errlen);

/* The grid spacing nl_hessian_stats would set, without the statistics pass (needed before NL_FIELD_FROB can be
   sampled for the bracket of nl_vesselness_spec). */
int nl_set_spacing(nl_ctx *ctx, const double spacing[3], char *err, size_t errlen);

/* nl_hessian_stats + nl_vesselness_step in ONE walk over the Hessian (filtering.py:555-585), for a mask threshold
   that is not known yet but predicted to satisfy  fsq_lo <= fsq_min <= fsq_hi  in units of frob_sq (the
   un-normalised squared Frobenius norm, filtering.py:538-543).  Voxels with frob_sq >= fsq_hi are treated as
   h_mask, below fsq_lo as not h_mask, the ones in between are parked.  Returns the statistics of
   nl_hessian_stats plus `overflow` (a queue region was too small).  Nothing is final until
   nl_vesselness_resolve confirms the bracket; if it cannot (or any_inf / overflow is set) the caller runs
   nl_vesselness_step as if this call had not happened.  Planes [z0, z1) must cover the owned planes.
   Availability: nl_ctx_info("vesselness_one_pass"). */
int nl_vesselness_spec(nl_ctx *ctx, const double spacing[3], float fsq_lo, float fsq_hi, int64_t z0, int64_t z1,
                       float *max_abs, float *max_frob_sq, int *any_inf, int *overflow, char *err, size_t errlen);

/* Device-resident threshold chain (csrc/chain.inc): the four data-dependent scalars of a scale -- gamma, the walk's bracket,
   max |H| and the Frobenius threshold (filtering.py:365-380, 421-444, 555-566) -- are computed by small kernels between the large
   ones, which read them from device memory, so a frame's scale loop is enqueued without a single wait:
     nl_chain_begin(n)            a frame of n scales (<= 16) starts
     nl_gauss_step ...            the cascade step of scale k, as always
     nl_chain_scale(...)          everything else of scale k: lattice histograms, thresholds, walk, exact histogram, resolve kernel
                                  (sz, sy, sx: lattice strides; division / margin / test_scale: frob_thresh_division, the relative
                                  half-width of the bracket, a factor on the predicted threshold (1.0; tests force misses with it))
     nl_chain_flush()             optional: start the download of the records now; kernels enqueued after it (the samples of the
                                  percentile threshold: nl_sample_gather_positive_begin) run while nl_chain_finish waits for the
                                  records alone and repeats the decisions on the host
     nl_chain_finish(...)         THE wait: per scale flags[k] (0 = stands), gamma, max |H|, the threshold, this context's h_mask count
   Every histogram and every derived scalar is logged, and nl_chain_finish repeats the arithmetic on the host with the code of
   the synchronous entry points and compares bit for bit: a difference, or any condition the fast path does not handle (no
   positive sample, degenerate triangle, +inf, queue overflow, bracket miss, empty scale), gives a non-zero flag and the caller
   redoes the frame with nl_sample_range_hist2 / nl_vesselness_spec / nl_vesselness_resolve.  Same results either way.
   nl_chain_log: test hook, the logged record of scale k (which: 0 Gaussian, 1 raw Frobenius, 2 normalised Frobenius samples). */
int nl_chain_begin(nl_ctx *ctx, int n_scales, char *err, size_t errlen);
int nl_chain_scale(nl_ctx *ctx, const double spacing[3], int64_t sz, int64_t sy, int64_t sx, double alpha_sq, double beta_sq,
                   double division, double margin, double test_scale, int64_t z0, int64_t z1, char *err, size_t errlen);
int nl_chain_flush(nl_ctx *ctx, char *err, size_t errlen);
int nl_chain_finish(nl_ctx *ctx, int *flags, double *gamma, double *max_abs, double *thr, int64_t *mask_count, char *err, size_t errlen);
int nl_chain_log(nl_ctx *ctx, int k, int which, int64_t *counts, float *edges, float *range, double *scalars, char *err, size_t errlen);

/* Second half of nl_vesselness_spec, with the arguments nl_vesselness_step takes (nl_set_frob_norm first).
   *hit = 1: the exact threshold lies inside the bracket; the parked voxels got the exact h_mask test and the scale
   is complete, bit-identical to nl_vesselness_step (`mask_count` as there).  *hit = 0: no effect. */
int nl_vesselness_resolve(nl_ctx *ctx, float gamma_sq, float alpha_sq, float beta_sq, int use_thr, float thr,
                          int *hit, int64_t *mask_count, char *err, size_t errlen);

/* The resolve kernel of nl_vesselness_resolve runs on a side stream so that the Gaussian cascade step of the next
   scale (which touches neither the vesselness volume nor the masks) can overlap it; every entry point that does
   touch them orders itself after it.  Passing mask_count = NULL to nl_vesselness_resolve keeps the call
   asynchronous; this returns the count later (and waits for the kernel). */
int nl_vesselness_count(nl_ctx *ctx, int64_t *mask_count, char *err, size_t errlen);

/* ---- 2-D images (im_info.no_z): a context of shape (1, ny, nx) switched to 2-D ------------------------------
   nl_set_ndim(ctx, 2) makes nl_hessian_stats, NL_FIELD_FROB, nl_vesselness_step and nl_mask_volume follow the
   reference's 2-D branches: Hessian (hxx, hxy, hyy) by np.gradient twice (filtering.py:461-490), closed-form
   float32 eigenvalues (filtering.py:675-690), the two-eigenvalue Frangi response (filtering.py:732-741) and the
   4-connected binary_opening.  nl_gauss_step is called with wz = NULL. */
int nl_set_ndim(nl_ctx *ctx, int ndim, char *err, size_t errlen);

/* One sigma of the multi-scale blob response (filtering.py:779-789): current = -gaussian_laplace(gauss, (s, s)) * s^2
   [* masks], running element-wise maximum.  wy2/wy0/wx2/wx0: scipy's `_gaussian_kernel1d` of order 2 / 0 for the Y and
   X axes, 2r+1 float64 weights each (truncate 4.0); s2 = float32(s**2); `gauss` is the current NL_FIELD_GAUSS
   (the reference's in-place cascade leaves the LAST scale's Gaussian in `frame`, filtering.py:811, 927-928). */
int nl_log2d_step(nl_ctx *ctx, const double *wy2, const double *wy0, const double *wx2, const double *wx0, int r,
                  float s2, int first, int use_mask, char *err, size_t errlen);

/* filtering.py:792-795 + 928-930: clip at 0, divide by (max + 1e-12) and by 10, NL_FIELD_FRANGI =
   maximum(NL_FIELD_FRANGI, blob).  Call after nl_filter_finish.  n_positive = pixels > 0 afterwards. */
int nl_log2d_finish(nl_ctx *ctx, int64_t *n_positive, char *err, size_t errlen);

/* vesselness * masks (filtering.py:926) -> NL_FIELD_FRANGI on planes [z0, z1) (-1, -1: owned).
   n_positive = number of OWNED voxels > 0 (the `sum > 0` test of filtering.py:1016-1017). */
int nl_filter_finish(nl_ctx *ctx, int64_t z0, int64_t z1, int64_t *n_positive, char *err, size_t errlen);

/* `_remove_edges` (filtering.py:969-1000, called at :931-932 when remove_edges=True) on the frame nl_filter_finish left
   on the device: in every Z plane (a 2-D image is one plane) the rows that hold a non-zero value span [rmin, rmax];
   min(margin, rmax - rmin + 1) rows at each end of the span are zeroed (margin = 15 in the reference).
   *n_positive = values > 0 that remain on the owned planes (`float(sum(frame)) > 0`, filtering.py:1014). */
int nl_remove_edges(nl_ctx *ctx, int margin, int64_t *n_positive, char *err, size_t errlen);

/* filtering.py:964-966: mask = frangi > thr; binary_opening (6-connected cross, one
   iteration, border 0); frangi *= mask. */
int nl_mask_volume(nl_ctx *ctx, float thr, char *err, size_t errlen);

/* nl_filter_finish + nl_mask_volume in one go for the common case (filtering.py:926 + 964-966), called INSTEAD of
   nl_filter_finish after the last scale with the percentile threshold of the NL_FIELD_VESSELNESS samples:
   the threshold pass reads the vesselness only where the cumulative mask has bits, the final pass reads it only where
   the opened mask has bits and writes zeros elsewhere (12 instead of 27 bytes per voxel of traffic).
   n_positive = OWNED voxels of vesselness * masks that are > 0 (what nl_filter_finish reports).  Requires at least
   one evaluated scale. */
int nl_mask_volume_fused(nl_ctx *ctx, float thr, int64_t *n_positive, char *err, size_t errlen);
/* The same epilogue with NO host decision in it (round 4; csrc/percentile.inc): nl_tail_enqueue gathers the positive lattice samples
   of `vesselness * masks` (strides sz, sy, sx: filtering.py:348-363), selects their q-th percentile on the device with numpy's
   float32 'linear' rule (filtering.py:963) -- on Z slabs from histograms all-reduced between the kernels, no sample leaves its
   rank --, and runs the percentile mask, the opening and the product (filtering.py:964-966) with the threshold read from device
   memory.  Enqueue only.  nl_tail_finish waits and reports (samples, the two order statistics a <= b, the interpolation weight,
   the threshold, the positive voxels of the product: global on a fused communicator); commit != 0 makes the result the frame
   (as nl_mask_volume_fused does) unless there was no positive sample -- the caller then takes the plain path. */
int nl_tail_enqueue(nl_ctx *ctx, int64_t sz, int64_t sy, int64_t sx, double q, char *err, size_t errlen);
int nl_tail_finish(nl_ctx *ctx, int commit, int64_t *n_samples, float *a, float *b, float *gamma, float *thr, int64_t *n_positive,
                   char *err, size_t errlen);
/* _mask_volume (filtering.py:952-967) on the finished Frangi frame with the percentile selected on the device (samples, selection,
   `frame > thr`, opening, product; one wait): the epilogue of 2-D images, of remove_edges runs and of slabs without the fused
   epilogue.  Reports (n, a, b, gamma, thr) for the host's check; n = 0: the frame was left as it is. */
int nl_mask_volume_dev(nl_ctx *ctx, int64_t sz, int64_t sy, int64_t sx, double q, int64_t *n_samples, float *a, float *b, float *gamma,
                       float *thr, char *err, size_t errlen);
/* tests: numpy.percentile(values, q) of n positive float32 values by the device's selection */
int nl_debug_percentile(nl_ctx *ctx, const float *values, int64_t n, double q, float *thr, float *a, float *b, char *err, size_t errlen);

/* D2H of NL_FIELD_FRANGI local planes [z0, z1) (filtering.py:1023-1031). */
int nl_filter_store(nl_ctx *ctx, float *host, int64_t z0, int64_t z1, char *err, size_t errlen);

/* Debug / test access: D2H of the current Gaussian volume, local planes [z0, z1). */
int nl_gauss_store(nl_ctx *ctx, float *host, int64_t z0, int64_t z1, char *err, size_t errlen);

/* ------------------------------------------------------------------ Z-slabs ----------- */
/* The reference has no distributed code; these entry points are new functionality: a
   volume too large for one device is cut into Z-slabs, one context per GPU, ghost planes exchanged
   between Z neighbours over RCCL/xGMI, scalars and histograms all-reduced. */

/* D2H / H2D of arbitrary local planes of a float field (NL_FIELD_GAUSS / NL_FIELD_FRANGI): the
   host-mediated exchange used by tests and by communicators without RCCL. */
int nl_planes_get(nl_ctx *ctx, int field, int64_t z0, int64_t z1, float *host, char *err, size_t errlen);
int nl_planes_put(nl_ctx *ctx, int field, int64_t z0, int64_t z1, const float *host, char *err, size_t errlen);

/* RCCL communicator: rank 0 calls nl_comm_unique_id (128 bytes), distributes it out of band, every rank
   calls nl_comm_init on its context. */
int nl_comm_unique_id(char *id128, char *err, size_t errlen);
int nl_comm_init(nl_ctx *ctx, int world, int rank, const char *id128, char *err, size_t errlen);
/* Loopback transport: an id from nl_comm_loopback_id makes nl_comm_init / nl_comm_init2 build a communicator between
   `world` contexts of THIS process (one host thread per rank, any devices of the process -- typically all on one GPU).
   Every exchange above -- ghost planes, bit planes, fused reductions, variable all-gathers -- then runs through the same
   code with the same pointers, offsets, counts, streams and events as over RCCL, the data moving by device-to-device
   copies on those streams: the N >= 2 paths are testable on a one-GPU box.  NELLIE_LOOPBACK_DELAY_US=n puts a random
   0..n us spin in front of every transfer (stress for missing stream dependencies). */
int nl_comm_loopback_id(char *id128, char *err, size_t errlen);

/* Exchange `depth` ghost planes of a float field with both Z neighbours (ncclSend/ncclRecv in one group,
   asynchronous on the context stream). */
int nl_halo_exchange(nl_ctx *ctx, int field, int64_t depth, char *err, size_t errlen);
/* The same for the `depth` owned planes that start `offset` planes inside the boundary (they land at the same distance
   from the interface on the other side): the per-step exchange of the cascade -- a rank computes every scale on its owned
   planes +- 4 and fetches, per step, only the r_z(s) planes beyond that from the neighbour that owns them.  async != 0
   (needs nl_comm_init2: a second communicator with its own unique id) runs the exchange on a stream and communicator of
   its own, ordered after the work submitted so far; the next nl_gauss_step waits for it, everything else runs beside it. */
int nl_halo_exchange_at(nl_ctx *ctx, int field, int64_t offset, int64_t depth, int async, char *err, size_t errlen);
int nl_comm_init2(nl_ctx *ctx, int world, int rank, const char *id128, char *err, size_t errlen);

/* All-reduce of a few host values through RCCL: dtype 0 = int64, 1 = float32; op 0 = sum, 1 = min, 2 = max. */
int nl_allreduce(nl_ctx *ctx, void *host_inout, int64_t count, int dtype, int op, char *err, size_t errlen);

/* ------------------------------------------------------------------ Label ------------- */

/* frangi_in_mem = xp.asarray(frangi_view) (labelling.py:547): host float32 -> NL_FIELD_FRANGI. */
int nl_label_load_frangi(nl_ctx *ctx, const float *host, int64_t z0, int64_t z1,
                         char *err, size_t errlen);

/* frangi *= (original > thresh) (labelling.py:550-552); the comparison is done in float64 on
   the exactly converted original, `thresh` already rounded by the host as numpy would. */
int nl_label_intensity_mask(nl_ctx *ctx, const void *host_original, int dtype, double thresh,
                            char *err, size_t errlen);
/* The same on planes [z0, z1) of the context's frame; `host_original` holds those planes of the original image (a Z slab masks
   the planes it owns: Label's intensity thresholds on a sharded frame). */
int nl_label_intensity_mask_planes(nl_ctx *ctx, const void *host_original, int dtype, double thresh, int64_t z0, int64_t z1,
                                   char *err, size_t errlen);

/* flat[offset::step] of a field (labelling.py:393, 412). */
int nl_flat_sample_gather(nl_ctx *ctx, int field, int64_t offset, int64_t step,
                          float *out, int64_t cap, int64_t *n, char *err, size_t errlen);

/* The same, positive samples only, compacted on the device (labelling.py:426-433); order unspecified. */
int nl_flat_sample_gather_positive(nl_ctx *ctx, int field, int64_t offset, int64_t step, float *out, int64_t cap,
                                   int64_t *n, char *err, size_t errlen);

/* labelling.py:467-509 on the device:
     mask = has_thr ? frangi > thr : 0; binary_fill_holes (6-conn) if fill_holes;
     label (26-conn); remove components with < min_area voxels; uniform_filter(3) > 0.5;
     label again.  Ids are int32, 1..K in raster order of each component's first voxel. */
int nl_label_run(nl_ctx *ctx, int has_thr, float thr, int64_t min_area, int fill_holes,
                 int64_t *n_labels, char *err, size_t errlen);

/* Label on Z-slabs.  The thresholded mask is 1 bit/voxel, so instead of stitching per-slab labellings
   every rank (1) packs `frangi > thr` of its OWN planes into a GLOBAL bit mask (nl_label_pack), (2) the bit
   planes are all-gathered (nl_label_bits_allgather over RCCL, or nl_label_bits_get/put through the host),
   (3) the run-level labelling of labelling.py:484-509 runs on the global mask on every rank
   (nl_label_run_global) and each rank paints only its own planes.  slab_plane0 has world+1 entries: the first
   global plane of every rank's slab, then gnz.  Rows are (global plane * ny + y). */
int nl_label_pack(nl_ctx *ctx, int has_thr, float thr, char *err, size_t errlen);
int nl_label_bits_get(nl_ctx *ctx, int64_t row0, int64_t nrows, uint64_t *host, char *err, size_t errlen);
int nl_label_bits_put(nl_ctx *ctx, int64_t row0, int64_t nrows, const uint64_t *host, char *err, size_t errlen);
int nl_label_bits_allgather(nl_ctx *ctx, const int64_t *slab_plane0, char *err, size_t errlen);
int nl_label_run_global(nl_ctx *ctx, int64_t min_area, int fill_holes, int64_t *n_labels, char *err, size_t errlen);

/* Label on Z-slabs WITHOUT replication (the production multi-GPU path; the reference's closest precedent is the chunk
   stitching of labelling.py:585-691 with its union-find :221-288, which however labels chunks independently and is not
   equivalent to the full-volume result -- this is).  Every rank labels its owned planes plus ONE ghost bit plane per
   interior side (exchanged with nl_slab_bits_exchange over RCCL, or nl_slab_bits_get / _put through the host).  A
   component that crosses an interface shows up on both ranks as a tree containing runs of the two planes both ranks see,
   and the k-th SEGMENT COMPONENT of such a plane (a connected piece of the plane's mask inside a band of rows: what the in-LDS
   level of the union-find joins) is the same voxels on both sides: `nl_slab_phase` hands those (tree, quantity) pairs -- a few
   hundred per plane -- to the host, which joins the trees of neighbouring ranks and patches the result back.  Three phases:
     phase 0 (fill)   6-connected background; quantity = 1 if the tree touches a face of the GLOBAL volume; after the
                      patch nl_slab_apply sets the enclosed background of the owned planes        (labelling.py:486)
     phase 1 (area)   26-connected foreground; quantity = voxels on the OWNED planes; after the patch (global sums)
                      nl_slab_apply(min_area) writes the kept-objects mask                        (labelling.py:489-501)
     nl_slab_majority majority filter of the kept-objects mask (its ghost planes exchanged first)  (labelling.py:503-505)
     phase 2 (number) 26-connected foreground; quantity = first run of the tree on the owned planes (INT32_MAX: none).
                      The owner of a component is the lowest rank holding voxels of it; nl_slab_number ranks the trees a
                      rank owns in raster order (and says which rank each tree of `set` got), nl_slab_paint adds the rank's
                      base id (exclusive sum of the lower ranks' counts) and takes the labels of trees owned elsewhere from
                      the host: ids 1..K in raster order of the first voxel, exactly scipy.ndimage.label's numbering of the
                      whole volume                                                                  (labelling.py:507)
   nl_slab_phase(phase, gather, block_ints, out, &need_ints, &nruns): one call per phase, one wait.  A rank's tables are a blob
   of int32: [n0 n1 n2 n3 | ints in the blob | overflow | runs | 0] [roots of plane 0..3] [values of plane 0..3], planes in the
   order ghost-low, first owned, last owned, ghost-high (n = 0 towards a side without a neighbour).  gather = 0: out receives
   this rank's blob (one block of block_ints); gather = 1 (needs nl_comm_init): the blobs of ALL ranks, all-gathered over RCCL
   on the context stream in fixed blocks (no size negotiation), rank r's at out + r * block_ints.  *need_ints: the largest blob;
   if it exceeds block_ints nothing was copied and the caller repeats with phase = -1 and a larger block (nothing is recomputed). */
int nl_slab_label_pack(nl_ctx *ctx, int has_thr, float thr, char *err, size_t errlen);
int nl_slab_bits_get(nl_ctx *ctx, int which, int64_t plane, uint64_t *host, char *err, size_t errlen);
int nl_slab_bits_put(nl_ctx *ctx, int which, int64_t plane, const uint64_t *host, char *err, size_t errlen);
int nl_slab_bits_exchange(nl_ctx *ctx, int which, char *err, size_t errlen);
int nl_slab_phase(nl_ctx *ctx, int phase, int gather, int64_t block_ints, int32_t *out, int64_t *need_ints, int64_t *nruns,
                  char *err, size_t errlen);
int nl_slab_patch(nl_ctx *ctx, int64_t n, const int32_t *roots, const int32_t *values, char *err, size_t errlen);
int nl_slab_apply(nl_ctx *ctx, int64_t min_area, char *err, size_t errlen);
int nl_slab_majority(nl_ctx *ctx, char *err, size_t errlen);
int nl_slab_number(nl_ctx *ctx, int64_t n_clear, const int32_t *clear, int64_t n_set, const int32_t *set, int64_t *n_local,
                   int32_t *ids_of_set, char *err, size_t errlen);
int nl_slab_paint(nl_ctx *ctx, int64_t base, int64_t n, const int32_t *roots, const int32_t *labels, char *err, size_t errlen);
/* Host only (no device, no context): joins the tables of `world` ranks (blobs as nl_slab_phase writes them, rank r's at
   blobs + r * block_ints).  One NODE per (rank, tree) that appears in a table, ranks in order, a rank's trees by ascending root;
   node_rank / node_root / node_val (int64, the tree's quantity) / node_comp receive up to `cap` nodes, *n_nodes their count,
   *n_comp the number of components after joining rank r's planes 2, 3 with rank r + 1's planes 0, 1 entry by entry; components
   are numbered in the order of their smallest node.  Returns NL_EINVAL if two ranks disagree about a shared plane.  Every rank
   calls it on the same gathered blobs and gets the same answer (the reference has no counterpart: SURVEY.md 8(e)). */
int nl_host_slab_join(int world, const int32_t *blobs, int64_t block_ints, int64_t cap, int64_t *n_nodes, int64_t *n_comp,
                      int64_t *node_rank, int32_t *node_root, int64_t *node_val, int64_t *node_comp, char *err, size_t errlen);

/* The positive samples of ALL ranks, compacted on the device, all-gathered over RCCL in fixed blocks and fetched, in one call with
   one wait: mode 0 = arr[::a, ::b, ::c] of `field` where > 0 (filtering.py:348-363, the samples of gamma / the percentile
   threshold), mode 1 = flat[a::b] where > 0 (labelling.py:418-433, Label's threshold).  block_items: a bound on the sample POINTS
   of any one rank, the same number on every rank (the callers derive it from the global shape).  out (cap floats) receives the
   samples rank by rank (order inside a rank unspecified: the consumers are order-free), counts[world] how many each rank gave. */
int nl_positive_samples_world(nl_ctx *ctx, int field, int mode, int64_t a, int64_t b, int64_t c, int64_t block_items,
                              float *out, int64_t cap, int64_t *counts, char *err, size_t errlen);
/* Variable-size all-gather of host bytes over RCCL (ncclAllGather on padded device staging): `recv` receives
   world * max_bytes bytes, rank r's block at r * max_bytes (its first bytes_of[r] bytes are valid; bytes_of has `world`
   entries and is filled here).  Used for the threshold samples and the slab tables, so that no data of the path
   travels through the control plane. */
int nl_allgather_bytes(nl_ctx *ctx, const void *send, int64_t nbytes, void *recv, int64_t max_bytes, int64_t *bytes_of,
                       char *err, size_t errlen);

/* The same when the caller does not know the largest block: the library gathers the sizes, uses their maximum (rounded up
   to 16) as the block size and lands the blocks in a page-locked buffer the CONTEXT owns: *recv (valid until the next
   call on this context), rank r's block at r * *stride. */
int nl_allgather_var(nl_ctx *ctx, const void *send, int64_t nbytes, void **recv, int64_t *stride, int64_t *bytes_of,
                     char *err, size_t errlen);

/* "Fused" reductions: with on != 0, nl_sample_minmax / nl_sample_hist / nl_sample_range_hist / nl_vesselness_spec return the
   value over ALL ranks of the communicator (range and counts of the lattice samples; max |H|, max frob_sq and the inf /
   overflow flags of the one-pass walk) -- the RCCL all-reduce runs on the context stream between the kernels, so the
   host-level nl_allreduce (one more device round trip each) is not needed for them.  The reference has no counterpart:
   its statistics are whole-volume numpy / cupy reductions (filtering.py:348-380, 555-562).  Requires nl_comm_init; every
   rank must issue the same sequence of calls. */
int nl_comm_fuse(nl_ctx *ctx, int on, char *err, size_t errlen);

/* D2H of the int32 label volume, local planes [z0, z1) (labelling.py:727-729). */
int nl_label_store(nl_ctx *ctx, int32_t *host, int64_t z0, int64_t z1, char *err, size_t errlen);

/* ------------------------------------------------------------------ frame streaming --- */
/* 3-D+T stacks: the reference moves every frame with a blocking xp.asarray / .get() (filtering.py:924, 1023;
   labelling.py:546, 727).  These entry points overlap the three legs instead: frame t+1 host->HBM on a copy
   stream (nl_input_load_async into slot 0/1, nl_input_select before nl_filter_begin), frame t computing, frame
   t-1's outputs HBM->host on a second copy stream (nl_outputs_stage: device copies into staging volumes;
   nl_outputs_fetch_async; nl_outputs_wait).  Host buffers must come from nl_pinned_alloc. */
int nl_pinned_alloc(void **ptr, int64_t bytes, char *err, size_t errlen);
int nl_pinned_free(void *ptr);
int nl_host_register(void *ptr, int64_t bytes, char *err, size_t errlen);   /* page-lock caller-owned memory */
int nl_host_unregister(void *ptr);
int nl_input_load_async(nl_ctx *ctx, int slot, const void *host_pinned, int dtype, char *err, size_t errlen);
int nl_input_select(nl_ctx *ctx, int slot, char *err, size_t errlen);
/* Blocks until the upload nl_input_load_async started into that slot has arrived.  The streamer with several lanes (contexts of one
   GPU working on different frames of a stack, nellie_amd/streaming.py) feeds them from ONE upload thread that keeps exactly one
   host -> HBM copy in flight: two concurrent copies share the link badly (measured: 2.49 ms alone, 3.57 ms each as a pair for a
   134 MB frame, profiles/r06_stream_lanes_trace.txt), and the upload is what bounds a float32 stack (filtering.py:917-924 is the
   blocking load this replaces). */
int nl_input_wait(nl_ctx *ctx, int slot, char *err, size_t errlen);
int nl_outputs_stage(nl_ctx *ctx, int with_labels, char *err, size_t errlen);
int nl_outputs_fetch_async(nl_ctx *ctx, float *frangi_pinned, int32_t *labels_pinned, char *err, size_t errlen);
int nl_outputs_wait(nl_ctx *ctx, char *err, size_t errlen);

/* Packed outputs.  Both products are ~98 % zeros; the dense download (8 B/voxel, the reference's blocking .get() per
   frame: filtering.py:1023, labelling.py:727) is what bounds a streamed stack.  nl_outputs_pack leaves in a staging buffer of
   the context, per volume: 1 bit per voxel (!= 0 / label > 0; rows padded to 64-voxel words), one u32 per row + 1 (index of
   the row's first item) and the items -- the non-zero float32 values in raster order, and ONE int32 label per maximal X-run
   of labelled voxels.  *nbytes = size of the blob; 0 = this frame does not pack (more than n / 4 items, or X-neighbours
   with different labels): fall back to nl_outputs_stage / nl_outputs_fetch_async.  nl_outputs_fetch_packed_async copies the
   blob to page-locked host memory on the download stream (nl_outputs_wait blocks until it landed; it is one of the
   copy-thread calls).  nl_outputs_unpack is host code: it expands a blob into dense (nz, ny, nx) arrays with `threads` host
   threads; dst_elems = elements each destination array holds -- a blob whose header describes another volume, or whose
   sections / row offsets do not fit its size, is refused before anything is written; zero_fill = 0 if the arrays are known
   to hold zeros (a freshly created file), rows without content are then not touched at all. */
int nl_outputs_pack(nl_ctx *ctx, int with_labels, int64_t *nbytes, char *err, size_t errlen);
/* on != 0: every nl_label_run of this context ends by enqueueing nl_outputs_pack(ctx, 1, ...) of its frame UNDER ITS OWN WAIT (the
   item counts stay on the device: no host round trip between counting and emitting); the nl_outputs_pack(ctx, 1, ...) that follows
   returns at once.  A streamed stack (labelling.py:701-734 has a blocking .get() per frame instead) loses two host waits per frame. */
int nl_outputs_pack_with_label(nl_ctx *ctx, int on, char *err, size_t errlen);
int nl_outputs_fetch_packed_async(nl_ctx *ctx, void *host_pinned, int64_t nbytes, char *err, size_t errlen);
int nl_outputs_unpack(const void *blob, int64_t nbytes, float *frangi, int32_t *labels, int64_t dst_elems, int zero_fill,
                      int threads, char *err, size_t errlen);
/* Host code: zero `bytes` bytes at `dst` with `threads` host threads.  For callers that unpack into arrays of their own (not
   freshly created sparse files): the fill of both dense outputs -- 8 B/voxel of host memory traffic, most of what an unpack
   with zero_fill = 1 takes -- can run on host threads WHILE the GPU still works on the frame, and the unpack (zero_fill = 0)
   then only scatters the non-zero items. */
int nl_host_zero(void *dst, int64_t bytes, int threads, char *err, size_t errlen);

/* ------------------------------------------------------------------ test hooks -------- */
/* Known-answer hook for the fused device routine (filtering.py:581-585 + 744-766): for n explicit
   Hessians h6[n][6] = (hxx,hxy,hxz,hyy,hyz,hzz) writes out4[n][4] = (l1,l2,l3 sorted by |.|, Frangi
   response).  impl 0 = production eigen-solve, 1 = libm acos/cos form. */
int nl_debug_eig_frangi(nl_ctx *ctx, const float *h6, int64_t n, int impl, float alpha_sq, float beta_sq,
                        float gamma_sq, float *out4, char *err, size_t errlen);

/* Introspection: "fast_div" (2 / 1 when the 2- / 3-instruction constant division was proven exact for the
   current spacings, 0: the float64 form), "hessian_tile_rows", "device_bytes", "gauss_yx_max_r" (largest in-plane
   radius whose Y and X passes share a kernel: see nl_gauss_step_ahead). */
int nl_ctx_info(nl_ctx *ctx, const char *key, double *value);

/* ------------------------------------------------------------------ timing ------------ */
/* HIP-event timing on the context stream (bench.py's roofline figures). */
int nl_timer_begin(nl_ctx *ctx, char *err, size_t errlen);
int nl_timer_end_ms(nl_ctx *ctx, float *ms, char *err, size_t errlen);
/* accumulated HIP-event time and launch count of one named kernel group since the last
   reset; names: "gauss", "hessian_stats", "vesselness", "sample", "finish", "mask_volume", "label" */
int nl_prof_enable(nl_ctx *ctx, int on);
int nl_prof_get(nl_ctx *ctx, const char *name, double *ms, int64_t *launches);
int nl_prof_reset(nl_ctx *ctx);

/* ---- Markers stage (nellie/segmentation/mocap_marking.py:648-703, use_im = 'distance', 3-D volumes) -----------------
   The stage after Label: distance transform of the labelled objects, their border shell, multi-scale LoG peaks of
   the distance image, intensity-based non-maximum suppression.  All three products are bit-exact.
   nl_markers_begin : mask = labels > 0 (mocap_marking.py:658-659); labels = NULL uses the labels nl_label_run left on
                      the device, intensity = NULL the resident input of nl_input_load (any dtype, cast to float32 as
                      `score_img[...] = intensity_im[...]` does, :595-596).
   nl_markers_distance : border = binary_dilation(mask) ^ mask; distance = float32(distance_transform_edt(mask)) clamped
                      at `clamp` = float32(2 * max_radius_px) (:440-448).  Exact: only background within floor(clamp)
                      voxels can matter.  n_mask (may be NULL) = object voxels.
   nl_markers_log_step : one sigma of :488-508 -- response = float32(-gaussian_laplace(distance, (s/z_ratio, s, s)) * s^2)
                      clamped at 0; a valid voxel (mask & distance > 0) whose response equals the maximum of its 3x3x3
                      neighbourhood (mode 'nearest') and beats every earlier scale becomes a peak.  Weights: scipy's
                      `_gaussian_kernel1d` of order 2 / 0, truncate 4.0, 2r+1 float64 values each, s2 = float32(s**2).
                      Radii 1 ... 63 on every axis (a 0.065 um pixel gives 21); in-plane radii above 12, or above ny, take
                      one-thread-per-voxel passes with scipy's multiple reflection (one more float32 volume, allocated then).
                      2-D image (context of shape (1, ny, nx)): wz2 = wz0 = NULL, sigma_vec = (s, s) (:323-324) -- the two
                      in-plane terms only; distance transform, border, 3x3 maxima and the suppression window are the
                      nz = 1 cases of the 3-D kernels.
   nl_markers_finish : :569-606 -- a peak survives if its float32 intensity is positive and equals the maximum over the
                      peaks within +-peak_min_distance.
   nl_markers_store  : marker (uint8 0/1), distance (float32), border (uint8 0/1); NULL pointers are skipped. */
int nl_markers_begin(nl_ctx *ctx, const int32_t *labels_host, const void *intensity_host, int dtype, char *err, size_t errlen);
int nl_markers_distance(nl_ctx *ctx, float clamp, int64_t *n_mask, char *err, size_t errlen);
/* use_im = 'frangi' (mocap_marking.py:675-679): the LoG of nl_markers_log_step runs on this float32 image (whole
   volume, host) instead of the distance image; NULL switches back.  Call between nl_markers_begin and the first step. */
int nl_markers_use_image(nl_ctx *ctx, const float *image, char *err, size_t errlen);
int nl_markers_log_step(nl_ctx *ctx, const double *wz2, const double *wz0, int rz, const double *wy2, const double *wy0,
                        const double *wx2, const double *wx0, int ryx, float s2, char *err, size_t errlen);
int nl_markers_finish(nl_ctx *ctx, int peak_min_distance, int64_t *n_markers, char *err, size_t errlen);
int nl_markers_store(nl_ctx *ctx, uint8_t *marker, float *distance, uint8_t *border, char *err, size_t errlen);

/* ---- Network stage, the two dense per-voxel steps (nellie/segmentation/networking.py; 3-D volumes and 2-D images) ----
   The skeletonisation itself (skimage, networking.py:394-409) and the per-object relabelling stay on the host.
   nl_skel_pixel_class  : `_get_pixel_class_impl` (networking.py:672-683) -- skel_mask = skel > 0;
                          class = min(4, convolve(skel_mask, ones(3,3,3), mode="constant", cval=0)) * skel_mask, uint8:
                          1 isolated, 2 tip, 3 edge, 4 junction.  n_skel (may be NULL) = skeleton voxels.
   nl_skel_branch_labels: `_get_branch_skel_labels` (networking.py:758-800) -- label((pc > 0) & (pc != 4),
                          structure = ones(3,3,3)): int32 ids 1..K in raster order of each component's first voxel.
                          pixel_class = NULL uses the classes the previous nl_skel_pixel_class left on the device.
   A context of shape (1, ny, nx) gives the reference's 2-D branch (ones(3,3)). */
int nl_skel_pixel_class(nl_ctx *ctx, const int32_t *skel, uint8_t *pixel_class, int64_t *n_skel, char *err, size_t errlen);
int nl_skel_branch_labels(nl_ctx *ctx, const uint8_t *pixel_class, int32_t *labels, int64_t *n_labels, char *err, size_t errlen);

#ifdef __cplusplus
}
#endif
#endif /* NELLIE_AMD_H */
