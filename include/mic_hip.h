/*
 * mic_hip.h -- C ABI of libmic_hip.so, the MI355X (gfx950) implementation of MIC's
 * parallel-strip encode/decode hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The
 * reference's Go package reaches it through cgo exactly as it reaches its own C codec
 * today (reference: ojph/mic_c.go:11-19); INTEGRATION.md shows the binding.
 *
 * Conventions (reference: ojph/mic_compress_c.h:26-38, ojph/mic_decompress_c.h:24-50,
 * ojph/mic_parallel.h:49-57):
 *   - the caller allocates every buffer; the library keeps no caller pointer after return;
 *   - 0 = success, negative = error class (MIC_ERR_*);
 *   - every entry point is thread-safe and may be called concurrently from any OS thread
 *     (reference: mic_parallel.h:47-48);
 *   - encoders take the caller's max_value: the Go API passes it
 *     (multiframecompress.go:15), the reference C derives it (mic_compress_c.c:774-775).
 *
 * There is no CPU fallback: every call fails with MIC_ERR_DEVICE when no gfx950 device
 * is usable.
 */
#ifndef MIC_HIP_H
#define MIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes --------------------------------------------------------------- */
#define MIC_OK                   0
#define MIC_ERR_ARGS            -1   /* bad pointer / dimension (mic_compress_c.c:918) */
#define MIC_ERR_NOMEM           -2
#define MIC_ERR_USE_RLE         -3   /* Go ErrUseRLE, fseu16.go:36; C: mic_compress_c.c:852 */
#define MIC_ERR_CAPACITY        -5   /* output buffer too small (mic_compress_c.c:865) */
#define MIC_ERR_CORRUPT         -6   /* malformed stream (mic_decompress_c.c:1004-1063) */
#define MIC_ERR_DEVICE          -7   /* HIP runtime / no gfx950 device */
#define MIC_ERR_INTERNAL        -8
#define MIC_ERR_UNSUPPORTED     -9
#define MIC_ERR_INCOMPRESSIBLE -10   /* Go ErrIncompressible, fseu16.go:33; C: -4 / -10 */

/* FSE flavour requested from an encoder: the entry of the reference's fallback chain
 * (multiframecompress.go:15-93).  2 -> CompressSingleFrame (2-state, then 1-state),
 * 4 -> CompressSingleFrame4State (4 -> 2 -> 1), 8 -> CompressSingleFrame8State. */
#define MIC_STATES_2 2
#define MIC_STATES_4 4
#define MIC_STATES_8 8

/* ---- library / device ------------------------------------------------------------- */
/* Selects the HIP device of the DEFAULT session, i.e. of every entry point below that takes host pointers (default 0).
 * Returns MIC_ERR_DEVICE when the device does not exist or is not gfx950.  A process that drives several GPUs -- the
 * reference's host is ONE process (goroutines, parallelstrips.go:77-93; re-entrancy: ojph/mic_parallel.h:47-48) -- creates
 * one session per device with mic_hip_session_create_on and uses the session entry points. */
int mic_hip_set_device(int device);
/* SEVERAL devices for the batch entry points below (mic_hip_compress_batch / _decompress_batch, mic_hip_pics_compress_batch /
 * _decompress_batch, mic_hip_mic2_compress / _decompress): a call's jobs are cut into one contiguous shard per listed device,
 * balanced by pixels -- the static assignment of the reference's fan-outs (parallelstrips.go:77-93, multiframecompress.go:186-209,
 * wsicompress.go:126-145) -- and the shards run side by side, each on a session of its device's pool with its own sub-batch
 * pipeline and transfer streams; results are written straight into the caller's buffers (no gather: the caller's memory is the
 * whole view).  This is how ONE host process -- the reference's host is one Go process -- gets past a single PCIe link: eight GPUs
 * are eight links.  devices[0] is also the device of every other host-pointer entry point; a device may be listed twice (two
 * shards on one GPU).  Waits for running calls; MIC_ERR_DEVICE when a device does not exist or is not gfx950.
 * mic_hip_get_devices: the current list (returns its length; at most cap entries are written). */
int mic_hip_set_devices(const int *devices, int n);
int mic_hip_get_devices(int *devices, int cap);
/* The cut those calls make, for callers that want to lay their work out to match it: n items of the given weights (pixels) into
 * `shards` contiguous shards -- item i belongs to shard k iff first[k] <= i < first[k + 1]; first has shards + 1 entries.
 * Needs no device. */
int mic_hip_shard_plan(const uint64_t *weights, int n, int shards, int *first);
/* "gfx950 <n CUs> ..." style description of the active device; "" if none. */
const char *mic_hip_device_name(void);
const char *mic_hip_version(void);

/* ---- unit codec: one frame / strip / plane ----------------------------------------- */
/* Replaces CompressSingleFrame{,4State,8State} (multiframecompress.go:15,38,67) and
 * mic_compress_{two,four,eight}_state (ojph/mic_compress_c.h:26-38).
 * out_cap >= MIC_HIP_FRAME_BOUND(width*height) is always sufficient: a frame whose every pixel escapes codes two tokens per
 * pixel, FSE only gives up (ErrIncompressible) at two bytes per token, and the NCount header of a 65536-symbol alphabet is < 128 KiB. */
#define MIC_HIP_FRAME_BOUND(npx) (4 * (size_t)(npx) + 135168)
/* The ONE capacity contract of every encoder below: a container's bound is the sum of its units' MIC_HIP_FRAME_BOUND plus its
 * header and table.  (The reference C allocates 2*w*h + 4096 per unit, ojph/mic_compress_c.c:918, and its Go wrapper 4*len + 4096,
 * ojph/mic_c.go:170; an all-escape frame needs the 4 bytes per pixel.)  Real frames code far below it: a caller that knows its
 * data may pass less and handle MIC_ERR_CAPACITY. */
#define MIC_HIP_PICS_BOUND(width, height, num_strips) \
    (20 + 8 * (size_t)(num_strips) + 4 * (size_t)(width) * (size_t)(height) + 135168 * (size_t)(num_strips))
#define MIC_HIP_MIC2_BOUND(width, height, nframes) \
    (20 + (size_t)(nframes) * (8 + MIC_HIP_FRAME_BOUND((size_t)(width) * (size_t)(height))))
int mic_hip_compress_frame(const uint16_t *pixels, int width, int height,
                           uint16_t max_value, int nstates,
                           uint8_t *out, size_t out_cap, size_t *out_len);

/* Replaces DecompressSingleFrame (multiframecompress.go:97) and
 * mic_decompress_{two,four,eight}_state (ojph/mic_decompress_c.h:24-50): the FSE flavour
 * (1/2/4/8-state, rANS-8) is auto-detected as in FSEDecompressU16Auto (fse2state.go:102). */
int mic_hip_decompress_frame(const uint8_t *compressed, size_t compressed_len,
                             uint16_t *pixels_out, int width, int height);

/* ---- bare FSE stage -------------------------------------------------------------------- */
/* Replaces FSECompressU16 (fsecompressu16.go:19), FSECompressU16TwoState (fse2state.go:22),
 * ...FourState (fse4state.go:24), ...EightState (fse8state.go:31) and RANSCompressU16EightState
 * (rans8state.go:31): flavour = 1, 2, 4, 8 or 108 (rANS-8).  No fallback chain: the sentinels
 * MIC_ERR_USE_RLE / MIC_ERR_INCOMPRESSIBLE come back exactly where the Go functions return
 * ErrUseRLE / ErrIncompressible.  out_cap >= 2*n + 135168 is always sufficient (past two bytes per symbol the encoders
 * return MIC_ERR_INCOMPRESSIBLE; the NCount header of a 65536-symbol alphabet is < 128 KiB). */
int mic_hip_fse_compress_u16(const uint16_t *symbols, size_t n, int flavour,
                             uint8_t *out, size_t out_cap, size_t *out_len);
/* The same with the caller's ScratchU16.TableLog (fseu16.go:101-102): the value optimalTableLog starts from
 * (fsecompressu16.go:480-518; 0 = the default 11, > 16 = MIC_ERR_ARGS like prepare(), fseu16.go:136-138).
 * ScratchU16.MaxSymbolValue (fseu16.go:98-99) has no counterpart: the reference only defaults it, nothing reads it. */
int mic_hip_fse_compress_u16_ex(const uint16_t *symbols, size_t n, int flavour, int table_log,
                                uint8_t *out, size_t out_cap, size_t *out_len);
/* Replaces FSEDecompressU16Auto (fse2state.go:102-116): magic-byte dispatch over all five
 * flavours.  *out_n receives the number of symbols written. */
int mic_hip_fse_decompress_u16_auto(const uint8_t *in, size_t in_len,
                                    uint16_t *out, size_t out_cap, size_t *out_n);
/* The same with the caller's ScratchU16.DecompressLimit (fseu16.go:87-91; 0 = the default 2 GiB - 1): MIC_ERR_CAPACITY exactly
 * where the reference returns "output size > DecompressLimit" -- it compares at every wrap of its 65536-symbol ring (and, for
 * 1-state streams, at the end), so an N-state stream of `count` symbols fails iff floor(count / 65536) * 65536 >= limit. */
int mic_hip_fse_decompress_u16_ex(const uint8_t *in, size_t in_len, int64_t decompress_limit,
                                  uint16_t *out, size_t out_cap, size_t *out_n);

/* ---- batch: many independent units in one call (one cgo crossing, one launch chain) --- */
/* Replaces the goroutine fan-out of parallelstrips.go:77-93 / :292-321, the frame loop of
 * multiframecompress.go:186-209 and the tile worker pool of wsicompress.go:126-145. */
typedef struct mic_hip_enc_job {
    const uint16_t *pixels;   /* in : width*height u16, row-major (host memory) */
    int32_t   width, height;  /* in  */
    uint16_t  max_value;      /* in  */
    uint16_t  nstates;        /* in : MIC_STATES_2/4/8 */
    uint8_t  *out;            /* in : caller buffer */
    size_t    out_cap;        /* in  */
    size_t    out_len;        /* out */
    int32_t   status;         /* out: MIC_OK or MIC_ERR_* for this unit */
    int32_t   nstates_used;   /* out: 8/4/2/1 flavour actually written */
} mic_hip_enc_job;

typedef struct mic_hip_dec_job {
    const uint8_t *compressed; /* in  (host memory) */
    size_t    compressed_len;  /* in  */
    uint16_t *pixels_out;      /* in : width*height u16 (host memory) */
    int32_t   width, height;   /* in  */
    int32_t   status;          /* out */
} mic_hip_dec_job;

/* Return value: MIC_OK when the batch ran (inspect per-job status), or a global error.
 * Host buffers are ordinary (pageable) memory or pinned memory (below); large batches run as a pipeline of sub-batches --
 * upload, kernels and download of neighbouring sub-batches overlap -- and concurrent callers run on different sessions of a
 * small pool (MIC_HIP_POOL sessions, default 3), as the reference's C codec runs concurrent goroutines (ojph/mic_parallel.h:47-48).
 * Environment, read once: MIC_HIP_WS_BUDGET_MB (device memory a default session may grow to), MIC_HIP_PIPELINE_PARTS (force the
 * number of sub-batches), MIC_HIP_TRACE=1 (the pipeline's stages with wall times on stderr). */
int mic_hip_compress_batch(mic_hip_enc_job *jobs, int njobs);
int mic_hip_decompress_batch(mic_hip_dec_job *jobs, int njobs);

/* Pinned host memory (hipHostMalloc) for a caller's frame and stream buffers: every entry point that takes host pointers
 * recognises such memory -- and memory the caller registered itself -- and DMAs it in place; ordinary memory (a Go slice) is staged
 * through pinned slots by the library's transfer threads (MIC_HIP_IO_THREADS, default half the host's cores, at most 8).
 * A cgo caller wraps the pointer with unsafe.Slice. */
void *mic_hip_host_alloc(size_t bytes);
void  mic_hip_host_free(void *p);

/* ---- PICS container --------------------------------------------------------------------- */
/* Replaces CompressParallelStrips{,4State,8State} (parallelstrips.go:55,128,199).
 * num_strips <= 0 is rejected with MIC_ERR_ARGS: the Go default (GOMAXPROCS) is a host
 * property and stays on the Go side.  out_cap >= MIC_HIP_PICS_BOUND(width, height, num_strips) is always sufficient. */
int mic_hip_pics_compress(const uint16_t *pixels, int width, int height,
                          uint16_t max_value, int num_strips, int nstates,
                          uint8_t *out, size_t out_cap, size_t *out_len);
/* The same with the index of the strip the error belongs to (the reference wraps it: "parallelstrips: strip %d: %w",
 * parallelstrips.go:97): *failed_strip = the first strip whose codec failed, -1 when the call succeeded or the error is not a
 * strip's (arguments, capacity of the header, the device). */
int mic_hip_pics_compress_ex(const uint16_t *pixels, int width, int height,
                             uint16_t max_value, int num_strips, int nstates,
                             uint8_t *out, size_t out_cap, size_t *out_len, int *failed_strip);
/* Header probe (parallelstrips.go:271-286). */
int mic_hip_pics_info(const uint8_t *compressed, size_t compressed_len,
                      int *width, int *height, int *num_strips, int *strip_height);
/* Replaces DecompressParallelStrips (parallelstrips.go:270) and mic_decompress_parallel
 * (ojph/mic_parallel.h:49-52; max_threads has no meaning on the GPU and is dropped).
 * width/height must equal the header's. */
int mic_hip_pics_decompress(const uint8_t *compressed, size_t compressed_len,
                            uint16_t *pixels_out, int width, int height);
/* ... and with the index of the strip that failed to decode ("parallelstrips: strip %d: %w", parallelstrips.go:316; -1: none). */
int mic_hip_pics_decompress_ex(const uint8_t *compressed, size_t compressed_len,
                               uint16_t *pixels_out, int width, int height, int *failed_strip);
/* Many images, one call: the strips of ALL jobs are one unit batch (the reference reaches the same parallelism by calling
 * CompressParallelStrips from many goroutines, each fanning out its strips: parallelstrips.go:77-93; a single image is eight
 * serial entropy chains and leaves the device idle).  Every job's file equals mic_hip_pics_compress's, byte for byte. */
typedef struct mic_hip_pics_enc_job {
    const uint16_t *pixels;   /* in : width*height u16 (host memory) */
    int32_t   width, height;  /* in  */
    uint16_t  max_value;      /* in  */
    uint16_t  nstates;        /* in : MIC_STATES_2/4/8 */
    int32_t   num_strips;     /* in : > 0 */
    uint8_t  *out;            /* in : caller buffer, out_cap >= MIC_HIP_PICS_BOUND(...) is always sufficient */
    size_t    out_cap;        /* in  */
    size_t    out_len;        /* out */
    int32_t   status;         /* out */
    int32_t   failed_strip;   /* out: the first strip whose codec failed (parallelstrips.go:97), -1: none / not a strip's error */
} mic_hip_pics_enc_job;
typedef struct mic_hip_pics_dec_job {
    const uint8_t *compressed; /* in : a PICS file (host memory) */
    size_t    compressed_len;  /* in  */
    uint16_t *pixels_out;      /* in : width*height u16 (host memory) */
    int32_t   width, height;   /* in : must equal the header's */
    int32_t   status;          /* out */
    int32_t   failed_strip;    /* out: the first strip that failed to decode (parallelstrips.go:316), -1: none */
} mic_hip_pics_dec_job;
int mic_hip_pics_compress_batch(mic_hip_pics_enc_job *jobs, int njobs);
int mic_hip_pics_decompress_batch(mic_hip_pics_dec_job *jobs, int njobs);

/* Replaces CompressSingleFrameGrad / DecompressSingleFrameGrad (multiframecompress.go:111-142): the unit codec with the
 * gradient-adaptive predictor (deltagradrlecompressu16.go) and the two-state -> one-state FSE chain. */
int mic_hip_compress_frame_grad(const uint16_t *pixels, int width, int height, uint16_t max_value,
                                uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_decompress_frame_grad(const uint8_t *compressed, size_t compressed_len,
                                  uint16_t *pixels_out, int width, int height);

/* ---- PICA container: content-adaptive strips, per-strip predictor choice ------------------------ */
/* Replaces CompressParallelStripsAdaptive (parallelstripsadaptive.go:54): strip boundaries by equal-cost partition of the rows'
 * summed |vertical delta| (adaptiveStripBoundaries, :222-289, float64 like the reference), every strip coded with both the avg
 * and the gradient-adaptive predictor (CompressSingleFrame / CompressSingleFrameGrad, two-state FSE) and the smaller kept, ties to
 * the gradient one (:97-105).  num_strips must be given (the reference's default is GOMAXPROCS). */
int mic_hip_pica_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int num_strips,
                          uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_pica_info(const uint8_t *compressed, size_t compressed_len, int *width, int *height, int *num_strips);
/* Replaces DecompressParallelStripsAdaptive (parallelstripsadaptive.go:141). */
int mic_hip_pica_decompress(const uint8_t *compressed, size_t compressed_len, uint16_t *pixels_out, int width, int height);

/* ---- MIC2 container, independent frames --------------------------------------------------- */
/* Replaces CompressMultiFrame(..., temporal=false) (multiframecompress.go:179) +
 * WriteMIC2 (multiframe.go:49).  frames = nframes*width*height u16, frame-major.  out_cap >= MIC_HIP_MIC2_BOUND(...) is always
 * sufficient (also for the temporal form below). */
int mic_hip_mic2_compress(const uint16_t *frames, int width, int height, int nframes,
                          uint16_t max_value,
                          uint8_t *out, size_t out_cap, size_t *out_len);
/* CompressMultiFrame(frames, w, h, maxValue, temporal=true) + WriteMIC2 (multiframecompress.go:179-224,
 * temporaldelta.go:11-23): frame 0 spatial, frame i > 0 = RLE + FSE(2-state, 1-state fallback) of
 * ZigZag(frame i - frame i-1).  Flags byte 0x03.  mic_hip_mic2_decompress reads both pipelines. */
int mic_hip_mic2_compress_temporal(const uint16_t *frames, int width, int height, int nframes,
                                   uint16_t max_value, uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_mic2_info(const uint8_t *compressed, size_t compressed_len,
                      int *width, int *height, int *nframes, int *temporal);
/* Replaces DecompressMultiFrame (multiframecompress.go:227) for independent-mode files. */
int mic_hip_mic2_decompress(const uint8_t *compressed, size_t compressed_len,
                            uint16_t *frames_out, size_t frames_cap_px);
/* DecompressFrame(data, frameIdx) (multiframecompress.go:266-315) with ExtractFrame (multiframe.go:131-142):
 * one frame of a MIC2 file; temporal files decode frames 0..frame_idx. */
int mic_hip_mic2_decompress_frame(const uint8_t *compressed, size_t compressed_len, int frame_idx,
                                  uint16_t *pixels_out, size_t pixels_cap);

/* ---- WaveletV2 -------------------------------------------------------------------------------- */
/* Replaces WaveletV2RLEFSECompressU16 and WaveletV2SIMDRLEFSECompressU16 (waveletfsecompressu16.go:303,
 * :374; identical streams): up to 8 levels of 5/3 integer lifting in Mallat layout, subband scan, zigzag
 * with 3-word escape, RLE with length prefix, 4-state FSE (no fallback), 11-byte header.
 * NOTE the argument order of the reference: (pixels, rows, cols, maxValue, levels). */
int mic_hip_wavelet_v2_compress(const uint16_t *pixels, int rows, int cols, uint16_t max_value, int levels,
                                uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_wavelet_v2_info(const uint8_t *compressed, size_t compressed_len,
                            int *rows, int *cols, int *max_value, int *levels);
/* Replaces WaveletV2RLEFSEDecompressU16 / WaveletV2SIMDRLEFSEDecompressU16 (:380, :493). */
int mic_hip_wavelet_v2_decompress(const uint8_t *compressed, size_t compressed_len,
                                  uint16_t *pixels_out, size_t out_cap_px);

/* Many frames of one shape side by side (the reference codes one image per call; a WaveletV2 file is ONE serial 4-state FSE
 * stream, so a single decode is one wave walking one chain -- the device pays off when frames are coded together).
 * compress_batch: frames = nframes x rows*cols u16, contiguous; frame i's file (byte-identical to the single call's) is written at
 * out + i*out_stride, out_lens[i] / status[i] per frame; a failing frame does not stop the others.
 * decompress_batch: nframes files of ONE shape (rows, cols, levels of files[0]; others: status MIC_ERR_ARGS); pixels_out receives
 * nframes x rows*cols u16. */
int mic_hip_wavelet_v2_compress_batch(const uint16_t *frames, int nframes, int rows, int cols, uint16_t max_value, int levels,
                                      uint8_t *out, size_t out_stride, size_t *out_lens, int32_t *status);
int mic_hip_wavelet_v2_decompress_batch(const uint8_t *const *files, const size_t *lens, int nframes,
                                        uint16_t *pixels_out, size_t out_cap_px, int32_t *status);

/* ---- MIC3 container: tiled RGB whole-slide images ---------------------------------------------- */
/* Replaces CompressWSI (wsicompress.go:27) + WriteMIC3 (wsiformat.go:99) for 8-bit RGB with the
 * YCoCg-R colour transform (forced on for RGB, wsiformat.go:93-95).  tile_w / tile_h = 0 select the
 * 256 x 256 default, levels <= 0 the automatic pyramid depth (wsiformat.go:273-285).  The pyramid
 * (2x2 box, wsipyramid.go:10-32), tile extraction, colour transform, plane statistics and every
 * plane's CompressSingleFrame run on the device; the host writes the container. */
int mic_hip_wsi_compress(const uint8_t *rgb, int width, int height, int tile_w, int tile_h, int levels,
                         uint8_t *out, size_t out_cap, size_t *out_len);
/* CompressWSI(pixels, width, height, channels, bitsPerSample, opts) with the reference's full signature: channels 3 /
 * 8 bits is the call above; channels 1 with 8 or 16 bits per sample (16-bit samples little-endian, bytesToUint16Slice,
 * wsicompress.go:573-587) is the greyscale path -- Downsample2xGrey (wsipyramid.go:34-55), one plane per tile and the tile blob is
 * that plane's blob (compressGreyTileBlob, :366-370), no colour-transform flag.  Other combinations: MIC_ERR_UNSUPPORTED. */
int mic_hip_wsi_compress_ex(const uint8_t *pixels, int width, int height, int channels, int bits_per_sample,
                            int tile_w, int tile_h, int levels, uint8_t *out, size_t out_cap, size_t *out_len);
/* WSIHeader.Channels / BitsPerSample / ColorTransform (wsiformat.go:169-227).  The decompress calls below write
 * channels * (bits_per_sample == 16 ? 2 : 1) bytes per pixel. */
int mic_hip_wsi_format(const uint8_t *compressed, size_t compressed_len, int *channels, int *bits_per_sample, int *color_transform);
/* ReadWSIHeader (wsicompress.go:299). */
int mic_hip_wsi_info(const uint8_t *compressed, size_t compressed_len, int *width, int *height,
                     int *tile_w, int *tile_h, int *levels, uint64_t *total_tiles);
int mic_hip_wsi_level_info(const uint8_t *compressed, size_t compressed_len, int level,
                           int *width, int *height, int *tiles_x, int *tiles_y);
/* Replaces DecompressWSITile (wsicompress.go:175): one tile, cropped at the level's edge;
 * *out_w x *out_h pixels are written (3 bytes each for RGB, 1 or 2 for greyscale). */
int mic_hip_wsi_decompress_tile(const uint8_t *compressed, size_t compressed_len, int level, int tile_x, int tile_y,
                                uint8_t *rgb_out, size_t out_cap, int *out_w, int *out_h);
/* All tiles of one pyramid level in a single batch, stitched into a level-sized image. */
int mic_hip_wsi_decompress_level(const uint8_t *compressed, size_t compressed_len, int level,
                                 uint8_t *rgb_out, size_t out_cap);
/* DecompressWSIRegion(data, level, x, y, w, h) (wsicompress.go:219-297): a rectangle of one pyramid level; w and h are
 * clamped to the level as the reference does and returned through out_w / out_h (may be NULL). */
int mic_hip_wsi_decompress_region(const uint8_t *compressed, size_t compressed_len, int level,
                                  int x, int y, int w, int h,
                                  uint8_t *rgb_out, size_t out_cap, int *out_w, int *out_h);

/* ---- single-frame RGB and the CLI's single-frame files ------------------------------------------ */
/* Replaces CompressRGB / DecompressRGB (rgbcompress.go:25-33): YCoCg-R, then the three planes as in a WSI tile blob
 * ([Y_len][Co_len][Cg_len] u32 LE + plane blobs); width and height travel out of band, as in the reference. */
int mic_hip_rgb_compress(const uint8_t *rgb, int width, int height, uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_rgb_decompress(const uint8_t *compressed, size_t compressed_len, int width, int height,
                           uint8_t *rgb_out, size_t out_cap);
/* MICR file (writeMICRFile, cmd/mic-compress/main.go:62-91): "MICR", width, height, CompressRGB blob. */
int mic_hip_micr_compress(const uint8_t *rgb, int width, int height, uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_micr_info(const uint8_t *compressed, size_t compressed_len, int *width, int *height);
int mic_hip_micr_decompress(const uint8_t *compressed, size_t compressed_len, uint8_t *rgb_out, size_t out_cap);
/* MIC1 file (writeMicFile, cmd/mic-compress/main.go:26-59): "MIC1", width, height, pipeline = 1, payload length,
 * CompressSingleFrame{,4State,8State} stream (nstates = 2, 4 or 8; the decoder auto-detects). */
int mic_hip_mic1_compress(const uint16_t *pixels, int width, int height, uint16_t max_value, int nstates,
                          uint8_t *out, size_t out_cap, size_t *out_len);
int mic_hip_mic1_info(const uint8_t *compressed, size_t compressed_len, int *width, int *height);
int mic_hip_mic1_decompress(const uint8_t *compressed, size_t compressed_len, uint16_t *pixels_out, size_t out_cap_px);

/* ---- device-resident sessions (inputs and outputs stay in HBM) ------------------------------ */
/* A session owns the workspace for up to max_units units of up to max_px pixels each and
 * runs the same kernels as the calls above on data that is already on the device.  This is
 * what bench.py times; it is also what a caller that produces / consumes pixels on the GPU
 * should use.  All pointers named d_* are device pointers. */
typedef struct mic_hip_session mic_hip_session;

int  mic_hip_session_create(mic_hip_session **s, int max_units, size_t max_px_per_unit);   /* on the default session's device */
/* The same on an explicit HIP device: the session's stream and workspace live there, and every call on the session makes that
 * device current for the calling thread first, so one host process can own a session per GPU and drive them from any thread. */
int  mic_hip_session_create_on(int device, mic_hip_session **s, int max_units, size_t max_px_per_unit);
int  mic_hip_session_device(mic_hip_session *s);
/* Device memory the session holds (bytes).  The unit codec lays its per-unit slabs out in two tiers: tier 1 -- one token per pixel
 * and an eighth, tables for 8192 symbols: about 7 bytes per pixel + 0.2 MB per unit -- is where every batch starts; a batch in
 * which a unit would cross a tier-1 capacity (more escapes than that, a 14-bit-and-up alphabet) is run again on the worst-case
 * slabs (42 bytes per pixel + 1.7 MB) and the session stays there: *tier2 (may be NULL) says so.  Results never depend on the tier. */
size_t mic_hip_session_workspace_bytes(mic_hip_session *s, int *tier2);
void mic_hip_session_destroy(mic_hip_session *s);

typedef struct mic_hip_unit {
    uint64_t px_offset;     /* first pixel of the unit, in u16 elements from d_pixels */
    int32_t  width, height;
    uint16_t max_value;
    uint16_t nstates;       /* 2 / 4 / 8, optionally | MIC_HIP_PRED_GRAD */
} mic_hip_unit;
/* OR'ed into mic_hip_unit.nstates: the unit uses the gradient-adaptive predictor of CompressSingleFrameGrad /
 * DecompressSingleFrameGrad (multiframecompress.go:111-142, deltagradrlecompressu16.go) instead of avg(left, top).  The
 * stream does not record its predictor (PICA keeps it in the strip's flags word), so decode units must carry it too. */
#define MIC_HIP_PRED_GRAD 0x200

/* Encode n units.  Compressed blobs are left in the session; *d_blobs receives the device
 * address of a packed buffer holding them back to back, h_offsets[n+1] (host) their byte
 * offsets, h_status[n] the per-unit status, h_nstates[n] (may be NULL) the flavour written.
 * Synchronous with respect to the host. */
int mic_hip_session_encode(mic_hip_session *s, const uint16_t *d_pixels,
                           const mic_hip_unit *units, int n,
                           const uint8_t **d_blobs, uint64_t *h_offsets,
                           int32_t *h_status, int32_t *h_nstates);
/* Decode n units whose compressed blobs are in d_blobs at h_offsets[i]..h_offsets[i+1];
 * pixels are written to d_pixels_out at units[i].px_offset. */
int mic_hip_session_decode(mic_hip_session *s, const uint8_t *d_blobs,
                           const uint64_t *h_offsets, const mic_hip_unit *units, int n,
                           uint16_t *d_pixels_out, int32_t *h_status);
/* Asynchronous forms used for timing: enqueue all kernels of a pass on the session's
 * stream (returned by mic_hip_session_stream as a hipStream_t) without touching the host;
 * results are fetched with the *_finish calls. */
void *mic_hip_session_stream(mic_hip_session *s);
/* The buffers a session hands out (*d_blobs, *d_streams) are reused by its next call; a caller that keeps them copies them out:
 * a device-to-device copy on the calling thread's current device, complete on return. */
int mic_hip_device_copy(void *d_dst, const void *d_src, size_t bytes);
int mic_hip_session_encode_enqueue(mic_hip_session *s, const uint16_t *d_pixels,
                                   const mic_hip_unit *units, int n);
int mic_hip_session_encode_finish(mic_hip_session *s, const uint8_t **d_blobs,
                                  uint64_t *h_offsets, int32_t *h_status, int32_t *h_nstates);
int mic_hip_session_decode_enqueue(mic_hip_session *s, const uint8_t *d_blobs,
                                   const uint64_t *h_offsets, const mic_hip_unit *units, int n,
                                   uint16_t *d_pixels_out);
int mic_hip_session_decode_finish(mic_hip_session *s, int32_t *h_status);
/* WaveletV2 on device-resident frames (BASELINE config 3 as bench.py times it): nframes frames of rows x cols u16, contiguous at
 * d_frames -> their streams WITHOUT the 11-byte file header (rows u32, cols u32, maxValue u16, levels u8,
 * waveletfsecompressu16.go:346-350 -- the caller holds those), packed back to back in the session: *d_streams,
 * h_offsets[nframes + 1], h_status[nframes]; *levels_applied = the level count the header carries (:321-330).  decode is the
 * inverse for streams of one shape and level count.  Files written from these streams equal mic_hip_wavelet_v2_compress's. */
int mic_hip_session_wavelet_v2_encode(mic_hip_session *s, const uint16_t *d_frames, int nframes, int rows, int cols, int levels,
                                      const uint8_t **d_streams, uint64_t *h_offsets, int32_t *h_status, int *levels_applied);
int mic_hip_session_wavelet_v2_decode(mic_hip_session *s, const uint8_t *d_streams, const uint64_t *h_offsets, int nframes,
                                      int rows, int cols, int levels, uint16_t *d_pixels_out, int32_t *h_status);
/* MIC3 on a device-resident slide (BASELINE config 5 as bench.py times it).  encode = CompressWSI (wsicompress.go:27-171) up
 * to, but without, the container: pyramid, tiles, YCoCg-R, plane modes and every plane's CompressSingleFrame on the device, the
 * coded planes kept in a store the session owns (device bytes + one host record per plane); *compressed_bytes = the size of the
 * MIC3 file they make.  write = WriteMIC3 (wsiformat.go:99-165) around the store: the file mic_hip_wsi_compress_ex would have
 * written, byte for byte.  decode_level = every tile of one pyramid level, from the store, into a device image of that level
 * (width * height * channels * bytes per sample); levels = the pyramid's shape. */
int mic_hip_session_wsi_encode(mic_hip_session *s, const uint8_t *d_pixels, int width, int height, int channels, int bits_per_sample,
                               int tile_w, int tile_h, int levels, uint64_t *total_tiles, uint64_t *compressed_bytes);
int mic_hip_session_wsi_write(mic_hip_session *s, uint8_t *out, size_t out_cap, size_t *out_len);
/* The same store as the container's PAYLOAD on the device -- every tile blob in container order, put together by a kernel:
 * *d_payload (valid until the session's next wsi call), its size, and tile_lens[cap >= total tiles] (host).  What a writer that
 * gathers several GPUs' tiles moves device to device; header, level table and tile index (wsiformat.go:99-165) are its own. */
int mic_hip_session_wsi_payload(mic_hip_session *s, const uint8_t **d_payload, uint64_t *payload_bytes, uint64_t *tile_lens, size_t cap);
int mic_hip_session_wsi_decode_level(mic_hip_session *s, int level, uint8_t *d_pixels_out, size_t out_cap);
int mic_hip_session_wsi_levels(mic_hip_session *s, int *levels, int *widths, int *heights, int cap);
/* Enables (1) / disables (0) per-kernel HIP-event timing of the enqueue calls. */
int mic_hip_session_set_timing(mic_hip_session *s, int enabled);
/* Per-kernel device time (ms, HIP events on the session stream) of the last enqueue:
 * names[i] / ms[i], returns the number of entries written (<= cap). */
int mic_hip_session_last_timings(mic_hip_session *s, const char **names, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif /* MIC_HIP_H */
