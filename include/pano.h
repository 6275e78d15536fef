/*
 * pano.h - C-ABI of libpano_hip.so: the MI355X (gfx950) implementation of the per-frame
 * panorama composition path of LeRoii/Img-Stitching.
 *
 * Drop-in boundary.  The reference's path is the header-only C++ class `ocvStitcher`
 * (reference include/ocvstitcher.hpp:254-1306) whose arithmetic lives behind five OpenCV
 * entry points: RotationWarper::warp / warpRoi, Blender::prepare / feed / blend.  Each entry
 * point below names the reference interface (file:line under the reference tree) it replaces.
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions, every call returns a pano_status (0 == RET_OK of
 *     reference include/stitcherglobal.h:13-14; PANO_ERR == RET_ERR == -1).
 *   - frames are 8-bit BGR interleaved (cv::Mat CV_8UC3), `stride` in bytes.
 *   - "d_" pointers are device (HBM) pointers, "h_" pointers are host pointers.
 *   - a ctx is single-caller; distinct ctxs share nothing (no globals, own buffers), so the
 *     reference's two-stitchers-on-two-threads pattern (src/master.cpp:314-318) needs no lock.
 *   - there is NO CPU fallback: compute entry points return PANO_ENODEVICE on a plan-only
 *     ctx (config.device < 0) and PANO_EHIP if the GPU runtime fails.
 */
#ifndef PANO_H
#define PANO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PANO_MAX_CAMS 8
#define PANO_MAX_BANDS 8

typedef struct pano_ctx pano_ctx;

typedef enum pano_status {
    PANO_OK = 0,          /* RET_OK  (stitcherglobal.h:13) */
    PANO_ERR = -1,        /* RET_ERR (stitcherglobal.h:14) */
    PANO_EINVAL = -2,     /* bad argument */
    PANO_ESTATE = -3,     /* call order (e.g. compose before prepare / masks) */
    PANO_EHIP = -4,       /* HIP runtime error, see pano_last_error() */
    PANO_ENODEVICE = -5,  /* compute call on a plan-only ctx */
    PANO_EWRAP = -6,      /* a camera's ROI straddles the +-pi seam and PANO_WRAP_IS_ERROR=1 asks to refuse it (README.md:27-29) */
    PANO_ENOMEM = -7
} pano_status;

enum { PANO_SPHERICAL = 0, PANO_CYLINDRICAL = 1 };   /* cv::SphericalWarper / cv::CylindricalWarper */
enum { PANO_BANDS_NO_BLEND = -1,                      /* Blender::NO (ocvstitcher.hpp:1190-1191) */
       PANO_BANDS_FROM_STRENGTH = -2 };               /* band rule of ocvstitcher.hpp:1188-1195 */

/* Mirrors stStitcherCfg (reference include/stitcherglobal.h:68-81) plus what ocvStitcher::init(yaml)
 * reads for the compose path (ocvstitcher.hpp:276-289): size, num_images, blendStrength, cut. */
typedef struct pano_config {
    int num_images;            /* stStitcherCfg.num_images, 1..PANO_MAX_CAMS */
    int width, height;         /* stStitcherCfg.width/height = yaml outPutWidth/outPutHeight */
    int projector;             /* PANO_SPHERICAL (reference default, ocvstitcher.hpp:1000) | PANO_CYLINDRICAL */
    float warped_image_scale;  /* ocvStitcher::warped_image_scale (last value of the 18N+1 list) */
    float blend_strength;      /* stStitcherCfg.blendStrength (yaml stitcherBlenderStrength) */
    int num_bands;             /* >=0 explicit MultiBandBlender::setNumBands; or PANO_BANDS_* */
    int cut[4];                /* m_cutParams x,y,w,h (cameras.yaml `cut`); w==0 -> whole panorama */
    int device;                /* HIP device ordinal; <0 = plan-only ctx (host geometry, no GPU) */
} pano_config;

/* ---- lifetime -------------------------------------------------------------------------- */
/* ocvStitcher::ocvStitcher() + init(yaml) (ocvstitcher.hpp:257, :262-358) */
pano_status pano_create(const pano_config* cfg, pano_ctx** out);
void pano_destroy(pano_ctx* ctx);
const char* pano_last_error(const pano_ctx* ctx);
const char* pano_version(void);

/* ---- camera parameters ------------------------------------------------------------------ */
/* camK[i] / cameraR[i], row-major f32 - useDefaultCamParams (ocvstitcher.hpp:423-450),
 * initCamParams (:452-520) */
pano_status pano_set_camera(pano_ctx* ctx, int i, const float K[9], const float R[9]);
/* Every way of setting cameras (this entry, the list, the file loaders) validates them: all 18 values finite, fx, fy > 0,
 * max |R^T R - I| <= 1e-3 and det R > 0; anything else is PANO_EINVAL with the reason in pano_last_error, and the context
 * keeps its previous cameras.  (The reference plans whatever it is given and fails later in a CV_Assert.) */
/* verifyCamParams (ocvstitcher.hpp:365-421): compare N ESTIMATED cameras (K_est / R_est: N x 9 row-major f32, e.g. from a
 * caller-side bundle adjustment) with the cameras the context holds (the defaults): per camera the Euclidean distance of the
 * Euler angles in degrees (rotationMatrixToEulerAngles, :229-253) against ex_thres (yaml stitcherCameraExThres) and of
 * (fx, fy) against in_thres (stitcherCameraInThres).  PANO_OK: the estimate is plausible; PANO_ERR (== RET_ERR): "environment
 * is not suitable for calibration, use default parameters" - *worst_camera (optional) names the first camera that failed.
 * Changes nothing in the context. */
pano_status pano_verify_cameras(pano_ctx* ctx, const float* K_est, const float* R_est, float ex_thres, float in_thres, int* worst_camera);
/* parse the reference's `18*N+1` comma-separated list (defaultCamParams, ocvstitcher.hpp:423-450;
 * cameras.yaml `cams:`), sets all N cameras and warped_image_scale */
pano_status pano_set_cameras_from_list(pano_ctx* ctx, const char* comma_separated_floats);
/* read the LAST record of a cameraparaout_<id>.txt log (initCamParams, ocvstitcher.hpp:452-520;
 * old 7-line shared-K format of 2222/cameraparaout_*.txt also accepted) */
pano_status pano_load_camera_file(pano_ctx* ctx, const char* path);

/* read back camK[i] / cameraR[i] / warped_image_scale - public members of ocvStitcher (ocvstitcher.hpp:1275-1299) that
 * callers and the loaders' tests look at; any of K, R, scale may be NULL */
pano_status pano_get_camera(const pano_ctx* ctx, int i, float K[9], float R[9], float* warped_image_scale);

/* append the current K / R / scale as a new record to a cameraparaout_<id>.txt log, in the format
 * saveCameraParams writes (ocvstitcher.hpp:522-562): "YYYY-MM-DD-HH-MM-SS:" / N lines of 18 values with trailing
 * commas / scale; values in ostream default formatting (6 significant digits) */
pano_status pano_save_camera_file(pano_ctx* ctx, const char* path);

/* ---- geometry: RotationWarper::warpRoi x N, resultRoi, band rule, Blender::prepare --------
 * (initSeam compose half, ocvstitcher.hpp:1054-1063, :1107-1121; per-frame :1186-1198).
 * Allocates every device buffer the ctx will ever use. */
pano_status pano_prepare(pano_ctx* ctx);
/* m_corners[i] / m_sizes[i] (ocvstitcher.hpp:1059-1060) */
pano_status pano_get_roi(const pano_ctx* ctx, int i, int xywh[4]);
/* resultRoi(m_corners, m_sizes) -> dst_sz (ocvstitcher.hpp:1110); xywh in warp coordinates */
pano_status pano_get_pano_rect(const pano_ctx* ctx, int xywh[4]);
/* MultiBandBlender::numBands() after prepare (cropped as blenders.cpp prepare does); -1 = Blender::NO */
pano_status pano_get_num_bands(const pano_ctx* ctx, int* num_bands);
/* bordered feed() tile of camera i in padded-canvas coordinates + copyMakeBorder widths
 * (MultiBandBlender::feed tl_new/br_new, top/bottom/left/right) */
pano_status pano_get_feed_tile(const pano_ctx* ctx, int i, int xywh[4], int tblr[4]);
/* m_cutParams (ocvstitcher.hpp:1210) */
pano_status pano_set_cut(pano_ctx* ctx, const int xywh[4]);
/* size of the image process() returns (cut applied) */
pano_status pano_get_output_size(const pano_ctx* ctx, int* w, int* h);

/* ---- blend masks: m_blenderMask[i] (ocvstitcher.hpp:1101, :1257) --------------------------- */
/* caller-supplied mask of camera i, ROI sized (pano_get_roi), host pointer */
pano_status pano_set_mask(pano_ctx* ctx, int i, const uint8_t* h_mask, int w, int h, size_t stride);
/* mask half of initSeam / updateMask (ocvstitcher.hpp:988-1101, :1218-1261) with the reference's
 * Voronoi seam option (src/stitching_detailed.cpp:728-729) instead of graph cut: seam-scale NEAREST
 * mask warp, VoronoiSeamFinder, dilate 3x3, resize INTER_LINEAR_EXACT, AND - all on the GPU */
pano_status pano_build_masks_voronoi(pano_ctx* ctx);
/* ocvStitcher::updateMask (ocvstitcher.hpp:1218-1261; the same steps inside initSeam, :981-1101) with the reference's
 * own seam finder, detail::GraphCutSeamFinder(COST_COLOR) (:1033-1035, :1244): the n stitcher-size BGR8 frames (host
 * pointers) are resized by seam_work_aspect (INTER_LINEAR_EXACT) and warped at the seam scale beside the NEAREST masks on
 * the GPU; per overlapping pair, in PairwiseSeamFinder::run order, the GPU builds the grid graph of findInPair (terminal
 * weights 10000, edge weights |a-b|^2 + |a'-b'|^2 + 1, +1000 at mask borders, gap 10), the host runs OpenCV's
 * Boykov-Kolmogorov max-flow (sequential by construction; csrc/pano_graphcut.hpp), the GPU applies the labels; then dilate
 * 3x3, resize INTER_LINEAR_EXACT, AND as in pano_build_masks_voronoi. */
pano_status pano_build_masks_graphcut(pano_ctx* ctx, const uint8_t* const* h_frames, const size_t* strides);
/* The same refresh BESIDE the frame loop.  ocvStitcher::process runs updateMask inline every 200 frames
 * (ocvstitcher.hpp:1152-1159), and the graph cuts cost several frame periods (77 ms for four 1080p cameras: the host
 * max-flow) - a capture loop at 60 fps loses frames there.  pano_refresh_masks_begin uploads the frames and warps them at
 * the seam scale on a stream of its own (a few ms; the caller's buffers are free again when it returns) and hands the graph
 * cuts to a thread of the library; pano_refresh_masks_poll returns at once and, the first time it finds the thread through,
 * installs the masks (*done = 1; the weights are rebuilt with the next frame, exactly as after pano_build_masks_graphcut:
 * same masks, bit for bit); pano_refresh_masks_wait blocks until then.  One refresh at a time (PANO_ESTATE otherwise);
 * pano_build_masks_graphcut and pano_destroy wait for a refresh under way; masks set by pano_set_mask / pano_build_masks_voronoi
 * while one runs are replaced when it is installed.  Call all three from the thread that composes. */
pano_status pano_refresh_masks_begin(pano_ctx* ctx, const uint8_t* const* h_frames, const size_t* strides);
pano_status pano_refresh_masks_poll(pano_ctx* ctx, int* done);
pano_status pano_refresh_masks_wait(pano_ctx* ctx);
pano_status pano_get_mask(pano_ctx* ctx, int i, uint8_t* h_mask, size_t stride);

/* ---- fused undistort front end (reference include/nvcam.hpp:823-833, :898-921, :1094) ----------------------
 * The reference undistorts each captured frame on the CPU before the stitcher sees it: resize(raw -> undist size),
 * remap(INTER_CUBIC) with initUndistortRectifyMap / getOptimalNewCameraMatrix(alpha=1) maps, crop `rect`,
 * resize(-> undist size), resize(-> outPut size).  With a front end set, pano_compose takes the RAW captured frames
 * (raw_w x raw_h) and the warp kernel samples them through the composed coordinate map of those five steps and the
 * projection - one bilinear tap set per panorama pixel, no intermediate images.  (A different resampling from the
 * reference's cubic + three bilinear passes: parity for this entry is defined against the fused map, DESIGN.md.)
 * Call before pano_prepare, for every camera or for none; all cameras share raw_w x raw_h <= 8192 x 8192. */
typedef struct pano_undistort {
    int raw_w, raw_h;        /* stCamCfg.camSrcWidth/Height: the frames handed to pano_compose */
    int undist_w, undist_h;  /* stCamCfg.undistoredWidth/Height */
    double K[9];             /* cameras.yaml `K` of the lens at undist size, row-major */
    double dist[4];          /* cameras.yaml `distorParams`: k1 k2 p1 p2 */
    int rect[4];             /* cameras.yaml `rect`: crop x y w h */
} pano_undistort;
pano_status pano_set_undistort(pano_ctx* ctx, int cam, const pano_undistort* u);
/* cv::getOptimalNewCameraMatrix(K, dist, undist size, 1) as the front end uses it (nvcam.hpp:830) */
pano_status pano_get_new_camera_matrix(const pano_ctx* ctx, int cam, double newK[9]);

/* ---- exposure: BlocksGainCompensator::apply (src/stitching_detailed.cpp:841) ---------------- */
/* block gain map of camera i (f32, gw x gh), bilinearly resized to the ROI like apply() does;
 * NULL removes it */
pano_status pano_set_gain_map(pano_ctx* ctx, int i, const float* h_gain, int gw, int gh);
/* ExposureCompensator::createDefault(GAIN_BLOCKS) + feed as ocvStitcher::initSeam runs it (ocvstitcher.hpp:981-1032;
 * CLI twin src/stitching_detailed.cpp:722-723): the n stitcher-size BGR8 frames (host pointers, like the cv::Mat the
 * reference holds at that point) are resized by seam_work_aspect (INTER_LINEAR_EXACT), warped at the seam scale
 * (INTER_LINEAR, BORDER_REFLECT) beside the INTER_NEAREST-warped masks, cut into block_w x block_h blocks
 * (BlocksGainCompensator's default is 32 x 32), and GainCompensator::feed's pairwise overlap statistics run on the GPU;
 * the normal equations are solved on the host in f64 with OpenCV's LU operation order and the two [1 2 1]/4 smoothing
 * passes follow.  The resulting maps are installed as by pano_set_gain_map, so the next pano_compose applies them. */
pano_status pano_estimate_gains(pano_ctx* ctx, const uint8_t* const* h_frames, const size_t* strides, int block_w,
                                int block_h);
/* the installed gain map of camera i (compensator->gains() / gain_maps_): *gw x *gh floats; h_gain may be NULL to ask
 * for the size only; 0 x 0 when camera i has none */
pano_status pano_get_gain_map(pano_ctx* ctx, int i, float* h_gain, int* gw, int* gh);

/* ---- per-frame path ---------------------------------------------------------------------- */
/* RotationWarper::warp(img, K, R, INTER_LINEAR, BORDER_REFLECT, dst) (ocvstitcher.hpp:1171):
 * writes the ROI-sized 8UC3 warped image of camera i.  Stage-level entry (parity tests, debugging);
 * pano_compose does not materialise this image. */
pano_status pano_warp(pano_ctx* ctx, int i, const uint8_t* d_src, size_t src_stride,
                      uint8_t* d_dst, size_t dst_stride, void* hip_stream);
/* RotationWarper::warp(mask255, K, R, INTER_NEAREST, BORDER_CONSTANT, dst) (ocvstitcher.hpp:1085) */
pano_status pano_warp_mask(pano_ctx* ctx, int i, uint8_t* d_dst, size_t dst_stride, void* hip_stream);

/* ocvStitcher::process(imgs, ret) (ocvstitcher.hpp:1141-1216) with device-resident frames:
 * N x (warp -> 16S -> feed) -> blend -> 8U -> cut.  Asynchronous on hip_stream. */
pano_status pano_compose(pano_ctx* ctx, const uint8_t* const* d_frames, const size_t* strides,
                         uint8_t* d_out, size_t out_stride, void* hip_stream);
/* the same with host cv::Mat-style buffers in and out (H2D, compose, D2H, synchronous): the exact
 * shape of process(vector<Mat>&, Mat&) */
pano_status pano_compose_host(pano_ctx* ctx, const uint8_t* const* h_frames, const size_t* strides,
                              uint8_t* h_out, size_t out_stride);

/* Page-locked host memory for frames and panoramas - what cv::cuda::HostMem(PAGE_LOCKED) is to a CUDA build of OpenCV.
 * pano_compose_host recognises page-locked buffers (these, hipHostMalloc, hipHostRegister) and DMAs them directly; any
 * other memory is staged through page-locked buffers of the ctx by a few copy threads (PANO_HOST_THREADS, default 8). */
void* pano_host_alloc(size_t bytes);
void pano_host_free(void* p);

/* The reference runs its two stitchers (upper / lower camera group) on two threads per frame
 * (src/master.cpp:314-318).  pano_compose_pair composes both in ONE launch sequence: a single warp launch over
 * all cameras of both contexts, one launch per pyramid / blend level for both canvases.  Results are identical to
 * two pano_compose calls; contexts whose level structure differs are simply composed one after the other. */
pano_status pano_compose_pair(pano_ctx* a, pano_ctx* b,
                              const uint8_t* const* d_frames_a, const size_t* strides_a, uint8_t* d_out_a, size_t out_stride_a,
                              const uint8_t* const* d_frames_b, const size_t* strides_b, uint8_t* d_out_b, size_t out_stride_b,
                              void* hip_stream);

/* ---- frames in flight --------------------------------------------------------------------------------------
 * The reference's loop composes one frame at a time (src/master.cpp:302-411: pop, process(), show).  One frame is a
 * chain of ten dependent launches, several of them far too small to fill the GPU, so the GPU idles between them.
 * Frame slots are extra sets of the per-frame buffers (camera pyramids, blend canvas; everything static - remap
 * tables, masks, weight pyramids - is shared): with n slots, frame k is composed into slot k % n on stream k % n and
 * the chains of consecutive frames overlap.  Results per frame are unchanged.
 *   pano_set_frame_slots(ctx, n)    after pano_prepare; 1 <= n <= PANO_MAX_FRAME_SLOTS; synchronises the device
 *   pano_select_frame_slot(ctx, k)  the slot the following feed / blend / compose calls work in (a host-side switch)
 * The caller keeps one stream per slot and does not reuse a slot before its previous frame is done (stream order
 * guarantees that when slot k always runs on stream k).  pano_compose_host and the pano_stream_* calls use slot 0.
 * Changing what the slots share - masks (pano_set_mask, pano_build_masks_*: the updateMask cadence of
 * ocvstitcher.hpp:1152-1159), gain maps, the cut - needs no synchronisation by the caller: the library lets the frames
 * in flight finish under the old state (a device-wide wait, once per change) before it rewrites it. */
#define PANO_MAX_FRAME_SLOTS 4
pano_status pano_set_frame_slots(pano_ctx* ctx, int n);
pano_status pano_select_frame_slot(pano_ctx* ctx, int k);
/* The streams to run the slots on: n hipStream_t (1 <= n <= PANO_MAX_FRAME_SLOTS) owned by the context, PROBED to run side by side.
 * The HIP runtime multiplexes all of a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) in an order the
 * caller does not control, and two slots' streams that land on one queue run their frames one after the other: measured on
 * config 2, 73.9 instead of 62.2 us per frame (docs/EXPERIMENTS.md, round 4).  The library creates candidates, times a pair of
 * one-wave spin kernels against every stream already taken and keeps the candidates that overlap; *distinct (optional) = how many
 * of the n sit on hardware queues of their own (fewer than n only when the runtime has fewer queues to give).  A few milliseconds,
 * once; call it while the device is otherwise idle.  Later calls hand out the same streams; a call that asks for MORE than the
 * context holds waits for the device, destroys the old ones and probes a new set (the handles of earlier calls are then dead).  The reference has no counterpart: it composes one frame at a time
 * (src/master.cpp:302-411). */
pano_status pano_frame_streams(pano_ctx* ctx, int n, void** streams, int* distinct);

/* ---- streaming form for a capture loop (BASELINE config 5: frames arrive in host memory at camera rate) ------
 * The reference's loop (src/master.cpp:302-411) pops one cv::Mat per camera from the capture queues and calls
 * process().  Here the library owns PINNED host buffers in PANO_STREAM_SLOTS slots: the producer writes camera i's
 * frame of slot s into pano_stream_input(s, i), pano_stream_submit(s) queues H2D -> compose -> D2H and returns at
 * once, pano_stream_wait(s) blocks until the panorama of slot s is in pano_stream_output(s).  The H2D of one slot
 * overlaps the kernels of the other (separate copy streams, events). */
#define PANO_STREAM_SLOTS 2
pano_status pano_stream_input(pano_ctx* ctx, int slot, int cam, uint8_t** h_ptr, size_t* stride);
pano_status pano_stream_output(pano_ctx* ctx, int slot, uint8_t** h_ptr, size_t* stride);
pano_status pano_stream_submit(pano_ctx* ctx, int slot);
pano_status pano_stream_wait(pano_ctx* ctx, int slot);

/* ---- camera-sharded (multi-GPU) form of the same path -------------------------------------- */
/* warp + Gaussian pyramid of the cameras selected by cam_bits (bit i = camera i) into the ctx's
 * pyramid slots; d_frames entries of unselected cameras are ignored */
pano_status pano_feed_cameras(pano_ctx* ctx, unsigned cam_bits, const uint8_t* const* d_frames,
                              const size_t* strides, void* hip_stream);
/* all pyramid slots are one allocation, slot i at base + i*slot_bytes (equal sized so that one
 * RCCL gather / all-gather lands every rank's slots in place) */
pano_status pano_get_pyramid_slots(pano_ctx* ctx, void** d_base, size_t* slot_bytes);
/* Blender::blend + 8U + cut over whatever the pyramid slots hold */
pano_status pano_blend(pano_ctx* ctx, uint8_t* d_out, size_t out_stride, void* hip_stream);

/* the same two with host buffers (a capture card per GPU host process): upload + feed, blend + download (synchronous), on the ctx's
 * own stream - the stream pano_gather_slots uses when it is given hip_stream == NULL */
pano_status pano_feed_cameras_host(pano_ctx* ctx, unsigned cam_bits, const uint8_t* const* h_frames, const size_t* strides);
pano_status pano_blend_host(pano_ctx* ctx, uint8_t* h_out, size_t out_stride);

/* The one exchange of the sharded path, over RCCL (one process per GPU): the pyramid slots of the cameras that OTHER ranks fed land
 * in this ctx's slots on `root` - ncclGroupStart, one ncclRecv (root) / ncclSend (owner) per run of consecutive slots with the same
 * owner, ncclGroupEnd, asynchronous on hip_stream.  This is the "single RCCL gather of the tiles onto rank 0" of the north star as
 * grouped point-to-point calls: the ranks of ANOTHER stitcher's cameras take no part, and the slots arrive in place.  The reference's
 * only inter-device transport is JPEG over UDP between two Jetsons (src/slave.cpp:88-145, src/panocamimpl.cpp:11-56).
 *   owner_rank[i]   the rank that ran pano_feed_cameras for camera i of this ctx (the same array on every rank)
 *   rccl_comm       an ncclComm_t that spans those ranks: the caller's own, or one made here -
 * pano_rccl_unique_id on one rank, the 128 bytes handed to the others by whatever channel the caller has, pano_rccl_comm_create on
 * every rank (collective).  librccl.so is loaded on first use; PANO_ENODEVICE when it is not there. */
#define PANO_RCCL_ID_BYTES 128
pano_status pano_rccl_unique_id(char id[PANO_RCCL_ID_BYTES]);
pano_status pano_rccl_comm_create(pano_ctx* ctx, const char id[PANO_RCCL_ID_BYTES], int world, int rank, void** rccl_comm);
pano_status pano_rccl_comm_destroy(void* rccl_comm);
pano_status pano_gather_slots(pano_ctx* ctx, void* rccl_comm, int rank, int root, const int* owner_rank, void* hip_stream);
/* What pano_gather_slots moves: by default not whole slots but the LIVE rectangles of every pyramid level of a camera (pano_get_live_rect:
 * what the blend on the root reads - on the 8 x 1080p rig 70 % of a slot), packed by a copy kernel into one message per owner and
 * unpacked in place on the root; every rank derives the same message sizes from the same masks.  PANO_GATHER_WHOLE_SLOTS=1
 * (environment, at pano_prepare) sends whole slots in place, as rounds 1 - 4 did.  packed_bytes_per_camera (optional, n values): the
 * bytes camera i contributes to a message; *slot_bytes (optional): a whole slot; *bytes_moved (optional): what this context has handed
 * to ncclSend / ncclRecv so far. */
pano_status pano_get_exchange_stats(pano_ctx* ctx, uint64_t* packed_bytes_per_camera, uint64_t* slot_bytes, uint64_t* bytes_moved);
/* ncclCommCount of a communicator (how many ranks RCCL itself says it spans), and the name the RCCL library was opened by
 * ("" when none could be).  Environment PANO_RCCL_LIB=<path>, read once at first use, names the library to open instead of the
 * system's librccl.so - a site build, or a test double that lets several ranks share one GPU (tests/src/fake_rccl.cpp). */
pano_status pano_rccl_comm_count(void* rccl_comm, int* ranks);
const char* pano_rccl_library(void);

/* ---- caller-side assembly of the two half panoramas (device buffers, BGR8) ------------------------------ */
/* src/master.cpp:321-326: cv::resize(up, up, down.size()) [INTER_LINEAR], cv::vconcat(up, down), black 10-row
 * divider centred on the seam.  d_out is down_w x 2*down_h.  `ctx` (either stitcher) names the device. */
pano_status pano_stack_master(pano_ctx* ctx, const uint8_t* d_up, int up_w, int up_h, size_t up_stride,
                              const uint8_t* d_down, int down_w, int down_h, size_t down_stride,
                              uint8_t* d_out, size_t out_stride, void* hip_stream);
/* src/panocamimpl.cpp:354-360: crop both halves to (min width) x (min height - 2*finalcut) from row `finalcut`,
 * vconcat, black 4-row divider.  d_out is min_w x 2*(min_h - 2*finalcut). */
pano_status pano_stack_finalcut(pano_ctx* ctx, const uint8_t* d_up, int up_w, int up_h, size_t up_stride,
                                const uint8_t* d_down, int down_w, int down_h, size_t down_stride, int finalcut,
                                uint8_t* d_out, size_t out_stride, void* hip_stream);

/* the same two on host cv::Mat-style buffers (what master.cpp / panocamimpl.cpp hold after process() returned): upload,
 * stack, download, synchronous.  h_out: down_w x 2*down_h, resp. min_w x 2*(min_h - 2*finalcut) */
pano_status pano_stack_master_host(pano_ctx* ctx, const uint8_t* h_up, int up_w, int up_h, size_t up_stride,
                                   const uint8_t* h_down, int down_w, int down_h, size_t down_stride,
                                   uint8_t* h_out, size_t out_stride);
pano_status pano_stack_finalcut_host(pano_ctx* ctx, const uint8_t* h_up, int up_w, int up_h, size_t up_stride,
                                     const uint8_t* h_down, int down_w, int down_h, size_t down_stride, int finalcut,
                                     uint8_t* h_out, size_t out_stride);

/* ---- measurement ------------------------------------------------------------------------- */
/* PANO_STAGE_BLEND0: the level-0 blend launch alone (the largest kernel of the blend stage), from the dispatch's own begin / end
 * timestamps like PANO_STAGE_WARP - the intervals rocprofv3 reports per kernel */
enum { PANO_STAGE_WARP = 0, PANO_STAGE_PYRAMID = 1, PANO_STAGE_BLEND = 2, PANO_STAGE_BLEND0 = 3, PANO_NUM_STAGES = 4 };
/* when enabled, hipEvents bracket each stage on the launch stream */
pano_status pano_set_profiling(pano_ctx* ctx, int enabled);
/* ms of each stage of the LAST compose (synchronises on its events) */
pano_status pano_get_stage_ms(pano_ctx* ctx, float ms[PANO_NUM_STAGES]);
/* totals since the last reset: the events live in a ring and are harvested lazily, so a timed loop
 * never waits for the GPU; launches[k] = number of stage-k intervals summed into total_ms[k] */
pano_status pano_get_stage_stats(pano_ctx* ctx, double total_ms[PANO_NUM_STAGES],
                                 uint64_t launches[PANO_NUM_STAGES], int reset);
/* algorithmic bytes of the warp kernel per compose: sum_cams (live share of W*H*3 read once + live part of Wt*Ht*3
 * written once: the bordered level-0 tile is stored as planar u8, the 8U->16S widening of ocvstitcher.hpp:1180
 * happens in registers downstream; "live" = the 64x16 blocks over pano_get_live_rect(level 0)) */
pano_status pano_get_warp_bytes(const pano_ctx* ctx, uint64_t* src_bytes, uint64_t* dst_bytes);
/* the warp's static remap table (the constant part of cv::detail::RotationWarper::buildMaps + cv::remap's index
 * arithmetic, ocvstitcher.hpp:1163): bytes one compose reads from it (2 per tile pixel in the packed form, 4 in the
 * dense form), the number of 64x16-pixel workgroups of the warp kernel, and how many of them hold pixels the packed
 * form cannot express (a BORDER_REFLECT fold inside a 4-pixel group, taps next to the last bytes of the frame) and
 * read the dense form with the per-pixel checked body.  Zeros when the warp projects on the fly. */
pano_status pano_get_warp_table_stats(const pano_ctx* ctx, uint64_t* table_bytes, uint64_t* blocks, uint64_t* blocks_checked);

/* Device-copy ceilings for the roofline ("also report against a measured device-copy ceiling"), measured by the library
 * itself with the launch and timing machinery of its kernels (per-launch begin / end events, the interval rocprofv3 reports):
 *   PANO_PROBE_COPY_F4        grid-stride copy, 16-byte loads and stores, four loads of a lane in flight; units = bytes to copy
 *   PANO_PROBE_COPY_F4_FLAT   the same bytes, one 16-byte element per lane, as many workgroups as it takes
 *   PANO_PROBE_COPY_K1_SHAPE  a copy with the warp kernel's traffic shape and none of its arithmetic: per 256-thread workgroup 4608 B of
 *                             source box by direct-to-LDS 16-byte loads + one 8-byte table entry per lane in, one dword per lane into
 *                             each of three planes out (6656 B in, 3072 B out); units = workgroups (config 2's warp launch: ~9300)
 * `sets` buffer sets are rotated through (sets x the bytes of a launch above the 256 MiB Infinity Cache: the cold figure; 1: warm),
 * `reps` timed launches follow max(2 sets, 4) untimed ones.  *GBps = bytes read + bytes written per launch / mean launch duration,
 * *us_per_launch that duration, *bytes_moved (optional) the numerator.  After pano_prepare (it runs on the context's stream).
 * Allocates and frees its own buffers; synchronises that stream.  The reference has no counterpart. */
enum { PANO_PROBE_COPY_F4 = 0, PANO_PROBE_COPY_K1_SHAPE = 1, PANO_PROBE_COPY_F4_FLAT = 2 };
pano_status pano_probe_copy(pano_ctx* ctx, int kind, uint64_t units, int sets, int reps, double* GBps, double* us_per_launch,
                            uint64_t* bytes_moved);
/* identity of the device code of this build: 16 hex digits of the SHA-256 over the kernel sources (csrc Makefile); the PMC traffic
 * figures under profiles/ name the id they were collected with */
const char* pano_kernel_source_id(void);

/* ---- stage inspection (parity tests) ------------------------------------------------------- */
/* The bytes of camera i's frame that the warp reads with the present masks: byte columns [rect[0], rect[0] + rect[2]) of rows
 * [rect[1], rect[1] + rect[3]), columns on 64-byte boundaries.  The remap table is static (fixed K / R), so every 64 x 16 patch of
 * the warp taps a fixed box of the frame, and the masks say which patches anything downstream reads: pano_compose_host,
 * pano_feed_cameras_host and pano_stream_submit upload this rectangle and nothing else (config 2: 70 % of a frame); a capture
 * pipeline that fills pano_stream_input buffers need not write the rest either.  Recomputed whenever the masks change.
 * Replaces nothing in the reference, which uploads whole frames inside every warp call (include/ocvstitcher.hpp:1171). */
pano_status pano_get_source_rect(const pano_ctx* ctx, int i, int rect[4]);
/* The part {x, y, w, h} of camera i's bordered tile at pyramid `level` that the library produces.
 * MultiBandBlender::feed (ocvstitcher.hpp:1202) weighs every Laplacian with the camera's weight pyramid, which is zero
 * away from the camera's blend mask; pixels of the tile that neither carry weight nor feed - through pyrDown / pyrUp -
 * a pixel that does are never read by anything, and the warp / pyramid kernels skip them (about a quarter of the tile
 * on the 8 x 1080p rig).  Follows the masks: the whole tile until the first compose after a mask change.  The
 * panorama is unaffected; pano_debug_get_level is defined inside this rectangle.  PANO_FULL_TILES=1 (environment, at
 * pano_prepare) produces whole tiles. */
pano_status pano_get_live_rect(const pano_ctx* ctx, int i, int level, int rect[4]);
/* A camera that straddles the +-pi seam of the projection gets, as from RotationWarper::warpRoi, a ROI as wide as the
 * whole u range, live at its two ends and dead in between (the reference avoids such cameras by stitching the ring as two
 * groups of four, README.md:27-29; here the ring may be one context).  gap = {first dead column, number of dead columns}
 * of the live rect at `level` ({0, 0}: none): the warp / pyramid kernels step over them. */
pano_status pano_get_live_gap(const pano_ctx* ctx, int i, int level, int gap[2]);
/* Gaussian level `level` of camera i's bordered tile, int16 x3 interleaved, tight rows */
pano_status pano_debug_get_level(pano_ctx* ctx, int i, int level, int16_t* h_dst, int* w, int* h);
/* For tests of the graph-cut seam finder: with a path set (nullptr / "" clears it), every later pano_build_masks_graphcut appends,
 * per overlapping pair in PairwiseSeamFinder::run order, the grid graph of GraphCutSeamFinder::Impl::findInPair exactly as the GPU
 * built it and the labels the max-flow gave it: int32 {i, j, W, H}, W*H f32 term (source - sink weight), wh (k <-> k + 1), wv
 * (k <-> k + W), W*H label bytes (1: source side = image i).  tests/test_gpu_parity.py checks on these graphs that the labelling is a
 * MINIMUM cut (its capacity equals an independent max-flow's value). */
pano_status pano_debug_graphcut_dump(pano_ctx* ctx, const char* path);
/* hipGraph replay of the frame's launch sequence (environment PANO_GRAPH=1 at pano_prepare; BASELINE config 5 "hipGraph capture"):
 * *graphs_held = graphs captured and kept (one per set of caller buffers and frame slot; -1: replay is off, or capture was not
 * available and the library launches directly), *replays = hipGraphLaunch calls so far.  For tests: proof that the graph path ran. */
pano_status pano_debug_graph_stats(const pano_ctx* ctx, int* graphs_held, uint64_t* replays);
/* f32 weight level of camera i (pyrDown chain of mask/255 with constant border) */
pano_status pano_debug_get_weights(pano_ctx* ctx, int i, int level, float* h_dst, int* w, int* h);
/* canvas: summed weights / collapsed image of a level */
pano_status pano_debug_get_canvas_weights(pano_ctx* ctx, int level, float* h_dst, int* w, int* h);
pano_status pano_debug_get_canvas(pano_ctx* ctx, int level, int16_t* h_dst, int* w, int* h);

#ifdef __cplusplus
}
#endif
#endif /* PANO_H */
