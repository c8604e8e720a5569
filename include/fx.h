/*
 * fx.h -- C ABI of libfx.so, the MI355X (gfx950) Hu-moment / PCA feature engine.
 *
 * The reference (mohammedazan/Image-Processing) is pure Python and has no FFI
 * of its own; the boundary it offers for this path is the function layer of
 * src/features/extract_2d_hu.py and src/features/extract_3d_pca_features.py.
 * Each entry point below names the reference function whose arithmetic it
 * replaces.  A Python host binds these with ctypes and keeps the reference
 * signatures above them.
 *
 * Conventions
 *   - every function returns 0 on success or a negative FX_E* code; the text of
 *     the last failure on the calling thread is fx_last_error();
 *   - all d_* pointers are caller-owned device pointers on the device given to
 *     fx_init (or the current HIP device); the library never frees or keeps
 *     them; optional outputs may be NULL;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the
 *     default stream) and the call returns without waiting for it;
 *   - no C++ exception crosses this boundary;
 *   - scratch memory that an entry point needs beyond its arguments is owned
 *     by the library per (device, stream): calls on different streams never
 *     share a buffer and may overlap freely; calls on one stream reuse it in
 *     stream order.  A buffer grows with the stream-ordered allocator
 *     (hipMallocAsync / hipFreeAsync on `stream`), so no entry point waits for
 *     the device unless its comment says so.  fx_release_stream returns a
 *     stream's buffers, fx_shutdown all of them.
 */
#ifndef FX_H
#define FX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_OK 0
#define FX_EINVAL (-22)   /* bad argument (shape, alignment, NULL output)    */
#define FX_ENOMEM (-12)   /* device scratch allocation failed                */
#define FX_EHIP (-5)      /* a HIP runtime call or kernel launch failed      */
#define FX_ENODEV (-19)   /* no usable gfx950 device                         */

/* Number of doubles per image in the moment record:
 * m00 m10 m01 m20 m11 m02 m30 m21 m12 m03, mu20 mu11 mu02 mu30 mu21 mu12 mu03,
 * nu20 nu11 nu02 nu30 nu21 nu12 nu03 -- the fields of cv2.moments(). */
#define FX_MOMENTS_PER_IMAGE 24

/* Selects `device` for the calling thread and creates the per-device state.
 * Counterpart of importing cv2/NumPy in the reference scripts; there is no
 * reference function.  Safe to call more than once. */
int fx_init(int device);

/* Releases every scratch buffer the library holds (waits for the devices). */
int fx_shutdown(void);

/* Returns the scratch buffers bound to `stream` on the current device, in
 * stream order (call it before destroying a stream the library has seen). */
int fx_release_stream(void *stream);

/* Message of the most recent failing call on this thread ("" if none). */
const char *fx_last_error(void);

/* ABI revision of this header. */
int fx_abi_version(void);

/*
 * Batched compute_hu_from_image (reference src/features/extract_2d_hu.py:443-478:
 * cv2.moments :449, cv2.HuMoments :450, -sign(h)*log10(|h|+1e-300) :471-478),
 * with the binarisation that precedes it on the silhouette path fused in
 * (`(mask > 127).astype(np.uint8) * 255`, extract_2d_hu.py:163,258,386,429).
 *
 *   d_img            B images of H x W uint8; row r of image b starts at
 *                    d_img + b*img_stride_bytes + r*pitch_bytes
 *   thresh           < 0: the pixel value is the weight (cv2.moments on a gray
 *                    image); 0..254: pixels > thresh weigh on_value, others 0
 *   on_value         1..255, used only when thresh >= 0
 *   d_moments24      optional, B x 24 doubles (cv2.moments record)
 *   d_hu7            optional, B x 7 doubles  (cv2.HuMoments, before the log)
 *   d_hulog7         optional, B x 7 floats   (the function's return value)
 *   n_bins, d_cpca   optional contour descriptor: when d_cpca != NULL, writes
 *                    B x (9+n_bins) floats = compute_pca_features
 *                    (extract_3d_pca_features.py:59-145) of the contour pixels
 *                    (foreground pixels, v > thresh or v > 127 when thresh < 0,
 *                    having a 4-neighbour that is background or outside the
 *                    image) taken as points (x, y, 0).  Rows of images with
 *                    fewer than 3 contour pixels are NaN (the reference
 *                    function raises ValueError there).  1 <= n_bins <= 64.
 *   d_ceig           optional with d_cpca: B x 6 doubles (lambda1..3, v1) in
 *                    float64 before the float32 cast.
 *
 * Raw moments are exact integers (the value cv2.moments accumulates), the
 * finalisation runs in IEEE double in OpenCV's operation order without FMA.
 * Limits: 1 <= W, H <= 16384.
 */
int fx_hu_batch_u8(const uint8_t *d_img, int64_t B, int H, int W,
                   int64_t pitch_bytes, int64_t img_stride_bytes,
                   int thresh, int on_value,
                   double *d_moments24, double *d_hu7, float *d_hulog7,
                   int n_bins, float *d_cpca, double *d_ceig,
                   void *stream);

/*
 * fx_hu_batch_u8 for callers that must not allocate: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  It has no
 * counterpart in the reference.  Same arguments, same kernels and bit-identical
 * results; the per-image sums, queue counters and redo list live in a block of
 * FX_HU_NOALLOC_MAX_IMAGES * 148 + 256 bytes (about 9.3 MiB) that is part of
 * the library's code object on every device, and the call makes no hipMalloc,
 * hipFree or other allocating call.  Limit: B <= FX_HU_NOALLOC_MAX_IMAGES for
 * images up to 1024 x 1024, B * ceil(W / 1024) * ceil(H / 256) <= the same
 * number for larger ones; beyond it the call fails with FX_EINVAL.  All calls
 * on one device share the block: calls that overlap in time on one device
 * (different streams or threads) race and must be ordered by the caller.
 * The block's address on a device is looked up by fx_init(device), else by
 * the first call on that device; a first call whose stream is being captured
 * fails with FX_EINVAL instead (run fx_init on the device beforehand).
 */
#define FX_HU_NOALLOC_MAX_IMAGES 65536
int fx_hu_batch_u8_noalloc(const uint8_t *d_img, int64_t B, int H, int W,
                           int64_t pitch_bytes, int64_t img_stride_bytes,
                           int thresh, int on_value,
                           double *d_moments24, double *d_hu7, float *d_hulog7,
                           int n_bins, float *d_cpca, double *d_ceig,
                           void *stream);

/*
 * Per-sample aggregation of the per-view rows (reference
 * src/features/extract_2d_hu.py:641-644: np.mean / np.std(ddof=0) over the
 * stacked float32 rows, concatenated to float32[14]), in NumPy's float32
 * evaluation order.  Rows of group g are d_hulog7[7*d_group_offsets[g]] ..
 * d_hulog7[7*d_group_offsets[g+1]-1]; d_group_offsets has G+1 entries on the
 * device.  d_out14 is G x 14 floats (mean[7] || std[7]); an empty group is NaN.
 */
int fx_hu_groupstat(const float *d_hulog7, const int64_t *d_group_offsets,
                    int64_t G, float *d_out14, void *stream);

/*
 * Batched compute_pca_features (reference
 * src/features/extract_3d_pca_features.py:59-145; the float32 file contents
 * are widened to float64 in-kernel as load_sample does at :48-56).
 *
 *   d_pts       points, `dims` floats each (2 or 3; z = 0 when dims == 2); may be
 *               NULL for a CSR batch whose sets are all empty (NaN rows)
 *   d_offsets   NULL: B sets of exactly N points, set b at point b*N;
 *               else B+1 point offsets on the device (CSR), and N = the number
 *               of points behind d_pts (any upper bound of d_offsets[B]): the
 *               work lists are then built on the device and the call does not
 *               wait for the stream.  N <= 0: unknown; the call copies the
 *               offsets to the host and waits for `stream` once before
 *               launching (and rejects decreasing offsets with FX_EINVAL,
 *               which the device path turns into NaN rows; with offsets that
 *               go back and forth, sets that together cover more points than
 *               N may get NaN rows too)
 *   n_bins      1..64 bins of the |projection on PC1| histogram
 *   normalize   != 0: divide centred points by their largest norm (:79-83)
 *   d_out       B x (9+n_bins) floats: lambda1..3, r1..3, ratio12, ratio23,
 *               trace, histogram.  Sets with fewer than 3 points give a NaN
 *               row (the reference function raises ValueError, :72-73).
 *   d_eig       optional, B x 6 doubles (lambda1..3, v1) before the cast.
 */
int fx_pca_batch_f32(const float *d_pts, const int64_t *d_offsets, int64_t B,
                     int64_t N, int dims, int n_bins, int normalize,
                     float *d_out, double *d_eig, void *stream);

/*
 * compute_pca_features on float64 coordinates (reference
 * src/features/extract_3d_pca_features.py:48-56 + :59-145: load_sample hands
 * over `arr[:, :3].astype(np.float64)` of whatever the file holds, and the
 * descriptor is computed from those values).  Same arguments and results as
 * fx_pca_batch_f32 with d_pts float64; for inputs that float32 does not hold
 * exactly (float64 files, arrays passed to compute_pca_features directly).
 * Every set takes the chunked path of the large clouds, whatever its size: a
 * correctness path, an order of magnitude below the float32 set kernels for
 * small sets.  Ragged batches need N > 0 (the number of points behind d_pts)
 * or cost one read-back of the last offset.
 */
int fx_pca_batch_f64(const double *d_pts, const int64_t *d_offsets, int64_t B, int64_t N,
                     int dims, int n_bins, int normalize,
                     float *d_out, double *d_eig, void *stream);

/*
 * Row-band form of fx_hu_batch_u8 for images that are split over several GPUs
 * (no reference counterpart: the reference handles one image per process; the
 * arithmetic is that of cv2.moments, src/features/extract_2d_hu.py:449).  Each
 * of the B images is a band of H rows that starts at row y_offset of a taller
 * image.  fx_hu_partial_u8 writes the band's exact integer raw moments m00,
 * m10, m01, m20, m11, m02, m30, m21, m12, m03 in the tall image's coordinates
 * as three 48-bit limbs each (value = l0 + l1 * 2^48 + l2 * 2^96), d_partial
 * int64 [B, 10, 3].  Limb arrays of the bands of one image are simply added
 * (an integer all-reduce over the ranks; up to 2^15 summands cannot overflow);
 * fx_hu_from_partial rounds the sums once to float64 and finalises them
 * exactly as fx_hu_batch_u8 does, so the result equals the unsplit image's.
 */
int fx_hu_partial_u8(const uint8_t *d_img, int64_t B, int H, int W,
                     int64_t pitch_bytes, int64_t img_stride_bytes, int thresh,
                     int on_value, int64_t y_offset, int64_t *d_partial,
                     void *stream);
int fx_hu_from_partial(const int64_t *d_partial, int64_t B, double *d_moments24,
                       double *d_hu7, float *d_hulog7, void *stream);

/*
 * Silhouette preprocessing of B gray views H x W (both <= 512), the step the
 * reference runs in front of the Hu moments.  Replaces, per view,
 *   compute_otsu        src/features/extract_2d_hu.py:129-188
 *   find_largest_bbox   :191-221
 *   preprocess_image    :352-440  (crop by the box + crop_margin, or the whole
 *                       view when preprocess_crop == 0; PIL nearest resize to
 *                       fit resolution x resolution; centred paste)
 *   clean_mask_np       :246-346  (opening, components below min_size pixels
 *                       dropped, the largest kept when none reaches it;
 *                       min_size < 0 selects the reference default
 *                       max(8, int(0.001 * resolution^2)))
 * and writes the cleaned 0/255 silhouette d_mask [B, resolution, resolution],
 * bit-identical to the reference's second return value of preprocess_image.
 * The reference has two branches, chosen by whether OpenCV is installed;
 * `mode` selects which one is reproduced:
 *   FX_SIL_OPENCV  the branch its production run took: cv2.GaussianBlur(5x5,
 *                  sigma 0) :139, cv2.threshold(THRESH_OTSU) :143, inversion
 *                  above half foreground :166-168, cv2.morphologyEx with the
 *                  5x5 MORPH_ELLIPSE, close x2 then open :173-175,
 *                  cv2.findContours(RETR_EXTERNAL) -> largest cv2.contourArea
 *                  -> cv2.boundingRect :200-221, 3x3 ellipse opening :262-263,
 *                  cv2.connectedComponentsWithStats(connectivity=8) :285-297;
 *   FX_SIL_PIL     the branch without OpenCV: PIL GaussianBlur(radius=1) :154,
 *                  NumPy Otsu :101-126, scipy.ndimage 3x3 closing / opening
 *                  :181-183, box of all non-zero pixels :209-214, 3x3 opening
 *                  :266, 4-connected scipy.ndimage.label :299-311.
 * crop_margin <= 1 is a fraction of the longer box side, > 1 is in pixels.
 * d_info (optional) int32 [B, 8]: Otsu level, 1 if the mask was inverted, crop
 * rectangle x0, y0, x1, y1 (exclusive ends), resized width and height.
 * The gray output of preprocess_image (PIL bicubic fit) is not produced here.
 */
#define FX_SIL_PIL 0
#define FX_SIL_OPENCV 1
int fx_silhouette_batch_u8(const uint8_t *d_img, int64_t B, int H, int W,
                           int64_t pitch, int64_t stride, int resolution,
                           double crop_margin, int preprocess_crop, int min_size,
                           int mode, uint8_t *d_mask, int32_t *d_info, void *stream);

/*
 * fx_silhouette_batch_u8 fused with the moment sums of the call that follows
 * it in the reference, compute_hu_from_mask / cv2.moments on the silhouette
 * (src/features/extract_2d_hu.py:443-470): d_partial int64 [B, 10, 3] receives
 * the exact integer raw moments m00, m10, m01, m20, m11, m02, m30, m21, m12, m03
 * of the 0 / 255 silhouette on its resolution x resolution canvas, as the
 * 48-bit limbs fx_hu_from_partial takes, summed from the row runs of the bit
 * mask while it is still on chip.  fx_hu_from_partial(d_partial, ...) then gives
 * bit for bit what fx_hu_batch_u8 gives on d_mask.  d_mask may be NULL (the
 * silhouettes are then never written: the reference keeps them only for its
 * previews); the other arguments are those of fx_silhouette_batch_u8.
 */
int fx_silhouette_moments_batch_u8(const uint8_t *d_img, int64_t B, int H, int W,
                                   int64_t pitch, int64_t stride, int resolution,
                                   double crop_margin, int preprocess_crop,
                                   int min_size, int mode, uint8_t *d_mask,
                                   int32_t *d_info, int64_t *d_partial,
                                   void *stream);

/*
 * Gray half of preprocess_image: pad_and_resize, src/features/extract_2d_hu.py
 * :224-240 (PIL ImageOps.contain with its default bicubic filter, then a
 * centred paste on a canvas of the background colour), applied to the crop
 * rectangle of each view.  d_info is the int32 [B, 8] block written by
 * fx_silhouette_batch_u8 (columns 2..5 = x0, y0, x1, y1) or NULL for the whole
 * view.  Bit-identical to PIL's 8-bit resampler (float64 weights, 22-bit fixed
 * point, a pass along x, then along y).  bg_value: 255 for "white", 0 for
 * "black".  d_out uint8 [B, resolution, resolution].  Views and resolution up
 * to 512 x 512.
 */
int fx_gray_fit_batch_u8(const uint8_t *d_img, int64_t B, int H, int W,
                         int64_t pitch, int64_t stride, const int32_t *d_info,
                         int resolution, int bg_value, uint8_t *d_out, void *stream);

/*
 * Point-cloud normalisation and farthest point sampling, the stage that
 * produces the fixed-size sets fx_pca_batch_f32 reads.  B clouds stored back to
 * back, cloud b = rows h_offsets[b] .. h_offsets[b+1] (HOST array of B + 1
 * entries).  The host arrays are copied through a page-locked block of the
 * stream before the call returns, which it does without waiting for `stream`:
 * the caller may reuse them at once, and outputs are valid in stream order.
 *
 * fx_center_scale replaces center_and_scale, src/preprocessing/prepare_all.py
 * :199-217: centroid = np.mean(points, axis=0) in float64 (rows added in
 * order), points minus centroid divided by the largest np.linalg.norm (left
 * undivided when that is 0), cast to float32.  d_pts is float32 (wide = 0) or
 * float64 (wide = 1) [total, 3]; d_out float32 [total, 3]; d_centroid float32
 * [B, 3]; d_maxdist (optional) float64 [B].  Bit-identical to NumPy.
 *
 * fx_fps_f32 replaces the sampling loop of farthest_point_sampling, :276-293,
 * for clouds with more points than n_samples (the reference's other branch,
 * tiling + RandomState.shuffle, stays with the caller, as does the draw of the
 * start index): float64 distances sqrt((dx*dx + dy*dy) + dz*dz) to the points
 * chosen so far, np.minimum, np.argmax with the first of equal maxima.
 * h_start (HOST, [B]) is the first index of each cloud.  d_idx int64
 * [B, n_samples]; d_sampled (optional) float32 [B, n_samples, 3] = the chosen
 * points.  Indices equal NumPy's.
 */
int fx_center_scale(const void *d_pts, int wide, const int64_t *h_offsets,
                    int64_t B, float *d_out, float *d_centroid, double *d_maxdist,
                    void *stream);
int fx_fps_f32(const float *d_pts, const int64_t *h_offsets, int64_t B,
               int n_samples, const int64_t *h_start, int64_t *d_idx,
               float *d_sampled, void *stream);

/*
 * The exchange step of the sharded path, for callers that hold an RCCL
 * communicator (ncclComm_t) of their own instead of torch.distributed: an
 * all-gather of every rank's [local_rows, row_floats] float32 feature block
 * into d_global [world * local_rows, row_floats], rank-major, enqueued on
 * `stream`.  Every rank passes the same local_rows (pad the last shard).  The
 * reference has no counterpart: its pool workers return rows to the parent
 * process (extract_2d_hu.py:746-760).  RCCL is resolved at the first call.
 */
int fx_allgather_features(const void *d_local, int64_t local_rows, int row_floats,
                          void *d_global, void *rccl_comm, void *stream);

/*
 * Host file I/O on native threads (no device work).  Per-file status codes:
 */
#define FX_IO_OK 0
#define FX_IO_EOPEN 1    /* cannot open / create */
#define FX_IO_EREAD 2    /* short read or write */
#define FX_IO_EFORMAT 3  /* not an .npy file, or a header this reader does not parse */
#define FX_IO_EDTYPE 4   /* a sample type or layout this reader leaves to NumPy / PIL */
#define FX_IO_ESHAPE 5   /* not 2-D with at least 3 columns */
#define FX_IO_EROWS 6    /* readable, but not of the batch's common size (the found size is reported) */
#define FX_IO_EWIDE 7    /* float64 samples that float32 does not hold exactly: to be evaluated as float64 */
#define FX_IO_ENOSPACE 8 /* fx_png_stage: the staging buffer is full */

/*
 * Batch form of load_sample, src/features/extract_3d_pca_features.py:48-56
 * (np.load + [:, :3]): reads n .npy files that each hold a [rows, >=3]
 * float32/float64 array and writes the first three columns as float32 into
 * out[i] (n x rows x 3, host memory, e.g. page-locked).  A float64 file is
 * taken over only when float32 holds all its samples exactly (load_sample
 * widens to float64 and the reference computes on those values), otherwise it
 * gets FX_IO_EWIDE and goes through fx_pca_batch_f64.  Only the header that
 * numpy.save writes is taken, character for character, and only when the file
 * holds the bytes its shape promises; a file that does not fit gets a status
 * and is left for the caller (np.load handles every other header, dtype and
 * layout and raises the reference's errors); status[i] / rows_found[i] per file.
 */
int fx_read_npy_points_f32(const char *const *paths, int64_t n, int64_t rows,
                           float *out, int32_t *status, int64_t *rows_found,
                           int n_threads);

/*
 * Batch form of save_descriptor, src/features/extract_3d_pca_features.py:148-158
 * (np.save of a 1-D float32 vector, allow_pickle=False): writes row i of
 * data (n x row_len float32, host memory) to paths[i], byte-identical to
 * np.save.
 */
int fx_write_npy_rows_f32(const char *const *paths, int64_t n, const float *data,
                          int64_t row_len, int32_t *status, int n_threads);

/*
 * Batch form of load_image, src/features/extract_2d_hu.py:84-98 (PIL branch:
 * Image.open(path).convert("L")): decodes n PNG files of H x W pixels into
 * out[i] (n x H x W uint8, host memory).  Handles what a renderer writes:
 * 8 bits per sample, gray / gray+alpha / RGB / RGBA, not interlaced, no tRNS;
 * gray = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16 as PIL, alpha ignored.
 * Any other file (other size: FX_IO_EROWS with found_hw[2i], found_hw[2i+1] =
 * its height and width; palette, 16 bit, interlaced, damaged, not a PNG) gets
 * a status and is left for the caller to hand to PIL, which decides what is an
 * error.
 */
int fx_decode_png_gray_u8(const char *const *paths, int64_t n, int H, int W,
                          uint8_t *out, int32_t *status, int32_t *found_hw,
                          int n_threads);

/*
 * The same decode with the inflate and the row filters on the device (no
 * counterpart in the reference, whose load_image decodes on the host; the
 * results are those of fx_decode_png_gray_u8 byte for byte).  Two steps:
 *
 * fx_png_stage (host threads) reads the files, checks the container exactly as
 * fx_decode_png_gray_u8 does (signature, chunk CRCs, IHDR: 8 bits per sample,
 * gray / gray+alpha / RGB / RGBA, not interlaced, no tRNS / PLTE, H x W) and
 * copies each file's zlib stream (its IDAT payloads joined) into `staging`
 * (host memory, page-locked if the copy to the device is to be asynchronous;
 * fx_png_stage_bytes gives a sufficient capacity): offset[i] (16-byte aligned,
 * at least 16 zero bytes behind the stream), length[i], channels[i], status[i]
 * (FX_IO_*; FX_IO_ENOSPACE when `capacity` is used up), found_hw as above;
 * *used = staging bytes written.
 *
 * fx_png_inflate_gray_u8 (device) takes the staged bytes in device memory with
 * the three arrays of the files to decode (device memory, n entries; files
 * whose stage status is not FX_IO_OK are left out by the caller) and writes
 * d_out[i] (n x H x W uint8) and d_status[i]: 0, or a non-zero code for a
 * stream it does not decode to exactly H rows of filter byte + W * channels
 * bytes with a matching Adler-32 (damaged data, a preset dictionary, a
 * filter type above 4): such a file is the caller's to hand to the host
 * decoder or PIL, like any other status.  W * channels <= 4096.  An entry
 * whose channel count is not 1 .. 4 (or does not fit such a row), whose
 * length is below 6 or above twice the bytes it is to produce + 1024 gets a
 * status too, and its stream is not read.
 */
int64_t fx_png_stage_bytes(const char *const *paths, int64_t n, int n_threads);
int fx_png_stage(const char *const *paths, int64_t n, int H, int W, uint8_t *staging, int64_t capacity,
                 int64_t *offset, int32_t *length, int32_t *channels, int32_t *status, int32_t *found_hw,
                 int64_t *used, int n_threads);
int fx_png_inflate_gray_u8(const uint8_t *d_staged, const int64_t *d_offset, const int32_t *d_length,
                           const int32_t *d_channels, int64_t n, int H, int W, uint8_t *d_out, int32_t *d_status,
                           void *stream);

/*
 * Per-column binarisation thresholds of a feature table: fit_thresholds,
 * src/models/train_bernoulli_nb.py:331-376 (col_thresh :348-356 = np.median /
 * np.quantile / _otsu_threshold_1d :307-328, its NumPy histogram branch).
 * d_x: rows x cols float64, row_stride elements between rows (device memory).
 * d_labels (or NULL) with `label` restricts the rows to one class, as the
 * "per_class" strategy does (:366-374); the mean over classes is the caller's.
 * method: 0 median, 1 quantile q (NumPy's default linear interpolation),
 * 2 Otsu.  constant_rule != 0: a column whose values all equal its first value
 * gets that value (:360-364).  d_thr[cols]; d_flags[cols] (optional): bit 0 =
 * constant column, bit 1 = the column holds NaN (its threshold is NaN, as
 * np.median gives); d_count[1] (optional): rows that took part.
 * Exact: order statistics by radix select, float64 arithmetic in NumPy's order.
 */
int fx_column_thresholds_f64(const double *d_x, int64_t rows, int cols,
                             int64_t row_stride, const int32_t *d_labels,
                             int label, int method, double q, int constant_rule,
                             double *d_thr, int32_t *d_flags, int64_t *d_count,
                             void *stream);

/*
 * apply_thresholds, src/models/train_bernoulli_nb.py:379-385:
 * d_out[r, c] = d_x[r, c] >= d_thr[c] as uint8 (rows x cols, dense).
 */
int fx_binarize_ge_f64(const double *d_x, int64_t rows, int cols,
                       int64_t row_stride, const double *d_thr, uint8_t *d_out,
                       void *stream);

/*
 * Renderer of the views the Hu path reads: render_fallback_matplotlib,
 * src/preprocessing/render_views.py:294-366 (the reference's renderer when
 * neither Open3D nor pyrender is installed), for B clouds x V camera poses.
 * d_pts: float32 [total, 3] with d_offsets[B + 1] (device).  d_cam: float64
 * [B, V, 9] = right, true_up (:312-320) and the look-at point of every pose;
 * d_scale[B] = (resolution * 0.35) / bbox diagonal (:329-333).  The kernel
 * projects (:323-335), applies the axis limits and equal-aspect box of
 * :353-355 on a resolution x resolution figure, snaps every marker to a pixel
 * as matplotlib's Agg backend does and blends the K x K coverage stamp of the
 * marker (host array, row-major, what Agg rasterises for the given point
 * size) in point order with Agg's integer blend onto a white canvas.
 * marker_value: 0 (bg_color "white") or 255 ("black": white markers; the
 * figure stays white because the axes patch is switched off, :344-349).
 * d_out: uint8 [B, V, resolution, resolution], the R = G = B channel of the
 * reference's image.  resolution <= 4096 (a view above 384 x 384 is drawn by several blocks, a band of rows each).
 *
 * fx_cloud_bounds_f32: per cloud min xyz, max xyz, mean xyz (float64 [B, 9])
 * as bbox_and_center :77-87 computes them from the float32 file values.
 */
int fx_cloud_bounds_f32(const float *d_pts, const int64_t *d_offsets, int64_t B,
                        double *d_bounds, void *stream);
/*
 * Camera table for fx_render_views_u8 on the device, so that a batch needs no
 * host round trip between fx_cloud_bounds_f32 and the renderer:
 * compute_camera_poses :93-164 (radius = radius_factor * bbox diagonal, camera
 * on the ring of view v, looking at the centroid, z up unless the view axis is
 * within 0.999 of it) and the in-plane basis of render_fallback_matplotlib
 * :306-320, per (cloud, view).  h_trig (host): V x 4 doubles = cos(elevation),
 * sin(elevation), cos(azimuth), sin(azimuth) of every view, computed by the
 * caller with the C library as the reference does (math.cos / math.sin; the
 * azimuth jitter comes from its RandomState).  V <= 64.
 * d_cam [B, V, 9], d_scale [B] = (resolution * 0.35) / bbox diagonal.
 */
int fx_camera_table_f64(const double *d_bounds, int64_t B, int V,
                        const double *h_trig, double radius_factor,
                        int resolution, double *d_cam, double *d_scale,
                        void *stream);
int fx_render_views_u8(const float *d_pts, const int64_t *d_offsets, int64_t B,
                       int V, const double *d_cam, const double *d_scale,
                       int resolution, const uint8_t *stamp, int K,
                       int marker_value, uint8_t *d_out, void *stream);

/*
 * Synthetic inputs for benchmarks and parity tests (no reference counterpart:
 * the reference reads rendered PNG views and sampled .npy clouds from disk).
 * Image `first_index + b` / set `first_index + b` depends only on (seed, that
 * index, the shape arguments), so any shard of a data set can be generated on
 * any rank.
 *
 * fx_synth_shapes_u8: H x W uint8 images with values {0, 255}; the foreground
 * is the union of 3..6 rotated ellipses and one 5..8-vertex polygon, rasterised
 * in integer arithmetic (an identical host generator exists for tests).
 * fx_synth_points_f32: N x 3 float32 Gaussian clouds, per-axis scale
 * log-uniform in [0.05, 1], random rotation.
 */
int fx_synth_shapes_u8(uint8_t *d_img, int64_t B, int H, int W,
                       int64_t pitch_bytes, int64_t img_stride_bytes,
                       uint64_t seed, int64_t first_index, void *stream);
int fx_synth_points_f32(float *d_pts, int64_t B, int64_t N, uint64_t seed,
                        int64_t first_index, void *stream);

/*
 * Timing helper for callers that launch on a raw HIP stream: average duration
 * in milliseconds of the kernels enqueued by the most recent fx_hu_batch_u8 /
 * fx_pca_batch_f32 call when profiling is switched on (fx_profile(1)): the
 * library then brackets that call's launches with HIP events on `stream`.
 * fx_profile_last_ms waits for the events.  Returns < 0 when nothing was
 * recorded.  A call that splits its units over several launches of its
 * dominant kernel (the silhouette stage runs large batches as two halves
 * side by side on two streams) brackets one launch per half and reports the
 * mean of the two durations: fx_profile_last_units is the number of units
 * one such launch processed, 0 when a single launch processed all the
 * call's units.  fx_profile_ms_back(k) is the same reading for the k-th
 * timed call before the most recent one (k = 0: the most recent; the last
 * 32 are kept, per host thread): a caller that reads a call's time only
 * after it has enqueued the following calls does not hold the host up.
 */
int fx_profile(int enable);
double fx_profile_last_ms(void);
double fx_profile_ms_back(int back);
long long fx_profile_last_units(void);

#ifdef __cplusplus
}
#endif
#endif /* FX_H */
