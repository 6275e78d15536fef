/*
 * pano_oracle.h - CPU restatement of the OpenCV-3.4 CPU semantics that
 * LeRoii/Img-Stitching's per-frame compose path (include/ocvstitcher.hpp:1141-1216,
 * src/stitching_detailed.cpp:778-904) executes through cv::detail::
 * {SphericalWarper, CylindricalWarper, MultiBandBlender, VoronoiSeamFinder,
 * BlocksGainCompensator}.
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it.  The product (img-stitching_amd/, include/pano.h)
 * never links, imports or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference ships no expected outputs and OpenCV (the
 * un-vendored dependency that carries the arithmetic, "opencv >= 3.4.0",
 * reference CMakeLists.txt:57-58, README.md:4-5) is not available in the build
 * container, so this restatement follows OpenCV 3.4's published algorithms
 * (modules/stitching/include/opencv2/stitching/detail/warpers_inl.hpp,
 * modules/stitching/src/{warpers,blenders,seam_finders,exposure_compensate}.cpp,
 * modules/imgproc/src/{imgwarp,pyramids,resize,morph,distransform}.cpp) and is
 * cross-checked only against an independent NumPy restatement (oracle/np_oracle.py)
 * and the ROI integers in SURVEY.md Appendix C.
 */
#ifndef PANO_ORACLE_H
#define PANO_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { PO_SPHERICAL = 0, PO_CYLINDRICAL = 1 };
enum { PO_INTER_NEAREST = 0, PO_INTER_LINEAR = 1 };
enum { PO_BORDER_CONSTANT = 0, PO_BORDER_REFLECT = 2, PO_BORDER_REFLECT_101 = 4 };

/* ---- A1: projectors (warpers_inl.hpp: ProjectorBase::setCameraParams,
 *          SphericalProjector / CylindricalProjector mapForward/mapBackward) */
typedef struct po_projector {
    int kind;
    float scale;
    float k[9], rinv[9], r_kinv[9], k_rinv[9];
} po_projector;

void po_projector_set(po_projector* p, int kind, float scale, const float K[9], const float R[9]);
void po_map_forward(const po_projector* p, float x, float y, float* u, float* v);
void po_map_backward(const po_projector* p, float u, float v, float* x, float* y);

/* ---- A2: ROI (warpers.cpp detectResultRoi / detectResultRoiByBorder, warpRoi; util.cpp resultRoi) */
void po_detect_result_roi(const po_projector* p, int src_w, int src_h, int tl[2], int br[2]);
void po_warp_roi(const po_projector* p, int src_w, int src_h, int rect_xywh[4]);
void po_result_roi(int n, const int* corners_xy, const int* sizes_wh, int rect_xywh[4]);

/* ---- A3: buildMaps + remap (imgwarp.cpp, 8U, INTER_LINEAR fixed point / INTER_NEAREST) */
void po_build_maps(const po_projector* p, int src_w, int src_h, float* xmap, float* ymap);
void po_remap_8u(const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                 const float* xmap, const float* ymap, int dw, int dh,
                 int interp, int border, uint8_t* dst, size_t dstride);
/* RotationWarper::warp: dst must hold warp_roi.w * warp_roi.h * cn bytes; returns tl in corner[2] */
void po_warp_8u(const po_projector* p, const uint8_t* src, int sw, int sh, size_t sstride, int cn,
                int interp, int border, uint8_t* dst, int corner[2]);

/* ---- A4: pyramids (pyramids.cpp pyrDown_/pyrUp_, BORDER_REFLECT_101 / default) */
void po_pyr_down_16s(const int16_t* src, int w, int h, int cn, int16_t* dst);
void po_pyr_down_32f(const float* src, int w, int h, float* dst);
/* which association of cv::pyrDown CV_32F's sum the restatement follows (see pano_oracle.c): vertical 0 scalar / 1 SSE2 or universal
 * intrinsics / 2 NEON, with a vector body of vbody floats; horizontal 0 scalar / 1 universal intrinsics / 2 the same with a fused
 * multiply-add, body hbody floats.  (0, *, 0, *) is the default and what the product computes. */
void po_set_pyrdown32f_variant(int vertical, int vbody, int horizontal, int hbody);
/* libm disagreement model for the projectors' sinf / cosf / atan2f / acosf: 0 none, 1 +1 ulp, 2 -1 ulp, 3 hashed -1 / 0 / +1 */
void po_set_trig_perturbation(int mode, unsigned seed);
void po_pyr_up_16s(const int16_t* src, int w, int h, int cn, int16_t* dst); /* dst is exactly 2w x 2h */

/* ---- misc imgproc used on the path */
void po_copy_make_border_16s(const int16_t* src, int w, int h, int cn, int top, int bottom, int left, int right,
                             int border, int16_t* dst);
void po_copy_make_border_32f(const float* src, int w, int h, int top, int bottom, int left, int right, float* dst);
void po_dilate3x3_8u(const uint8_t* src, int w, int h, uint8_t* dst);
/* resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR_EXACT): sampling grid from dsize/ssize (ocvstitcher.hpp:1099,1256) */
void po_resize_linear_exact_8u(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh);
/* resize(src, dst, Size(), fx, fy, INTER_LINEAR_EXACT): dsize = cvRound(ssize*f) but the sampling grid is 1/f
 * (cv::resize keeps inv_scale = fx when dsize is empty; ocvstitcher.hpp:988,1230).  fx, fy <= 0: same as above */
int po_resize_dsize(int ssize, double f);
void po_resize_linear_exact_8u_fxy(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh, double fx,
                                   double fy);
void po_distance_l1(const uint8_t* src, int w, int h, float* dst);

/* ---- A5: band rule of the callers (ocvstitcher.hpp:1188-1195).  Returns -1 for Blender::NO */
int po_bands_from_strength(int dst_w, int dst_h, float strength);

/* ---- A5: MultiBandBlender (blenders.cpp) */
typedef struct po_blender po_blender;
po_blender* po_blender_create(int num_bands /* >=0: MultiBand; -1: Blender::NO */);
void po_blender_destroy(po_blender* b);
void po_blender_prepare(po_blender* b, int n, const int* corners_xy, const int* sizes_wh);
void po_blender_feed(po_blender* b, const int16_t* img16sc3, const uint8_t* mask, int w, int h, int tl_x, int tl_y);
/* result is dst_roi_final sized (w*h*3 int16 + w*h uint8) */
void po_blender_blend(po_blender* b, int16_t* dst16sc3, uint8_t* dst_mask);
/* introspection for stage-level parity tests */
int po_blender_num_bands(const po_blender* b);
void po_blender_dst_roi(const po_blender* b, int rect_xywh[4]);        /* padded */
void po_blender_dst_roi_final(const po_blender* b, int rect_xywh[4]);
void po_blender_level_size(const po_blender* b, int level, int wh[2]);
const int16_t* po_blender_level_laplace(const po_blender* b, int level);
const float* po_blender_level_weights(const po_blender* b, int level);
/* geometry of the last feed(): bordered tile rect in canvas coords + border widths */
void po_blender_last_tile(const po_blender* b, int rect_xywh[4], int tblr[4]);

/* ---- A7: VoronoiSeamFinder::find (seam_finders.cpp) on u8 masks, in place */
void po_voronoi_find(int n, const int* corners_xy, const int* sizes_wh, uint8_t** masks);
/* detail::GraphCutSeamFinder(COST_COLOR)::find - the reference's seam finder (ocvstitcher.hpp:1033-1035, :1244):
 * PairwiseSeamFinder::run order, findInPair with GCGraph<float> (Boykov-Kolmogorov as OpenCV implements it).
 * images: the 8UC3 warps (dense); masks are updated in place */
void po_graphcut_find(int n, const int* corners_xy, const int* sizes_wh, const uint8_t* const* images, uint8_t** masks);
/* ocvStitcher::updateMask with the reference's own seam finder: frames -> m_blenderMask (each warp_roi(i) sized) */
void po_prepare_masks_graphcut(int n, int kind, int src_w, int src_h, const uint8_t* const* frames, const float* K9s,
                               const float* R9s, float scale, uint8_t** masks_out);
float po_gc_grid_max_flow(int W, int H, const float* term, const float* wh, const float* wv, uint8_t* labels);

/* ---- A8: BlocksGainCompensator::apply (exposure_compensate.cpp, 3.4) */
void po_resize_linear_32f(const float* src, int sw, int sh, float* dst, int dw, int dh);
void po_gain_apply_8uc3(uint8_t* img, int w, int h, const float* gain_map /* w*h */);

/* ---- mask preparation of ocvStitcher::initSeam / updateMask with a Voronoi seam finder
 *      (ocvstitcher.hpp:975-1101, 1218-1261; stitching_detailed.cpp:726-756, 838-850).
 * masks_out[i] must hold warp_roi(i).w * warp_roi(i).h bytes. */
void po_prepare_masks_voronoi(int n, int kind, int src_w, int src_h, const float* K9s, const float* R9s,
                              float warped_image_scale, uint8_t** masks_out);

/* ---- fused undistort front end (reference include/nvcam.hpp:823-833, :898-921, :1094).
 * The reference undistorts every captured frame on the CPU before the stitcher sees it:
 *   resize(raw -> undist size) -> remap(INTER_CUBIC, maps of initUndistortRectifyMap with
 *   getOptimalNewCameraMatrix(alpha=1)) -> crop rect -> resize(-> undist size) -> resize(-> outPut size)
 * The fused form composes the COORDINATE maps of those five steps behind the spherical map and samples the raw frame
 * once (fixed-point bilinear, like the warp).  It is a different resampling from the reference chain (one bilinear
 * tap set instead of cubic + three bilinear passes), so its parity is defined against this restatement only. */
typedef struct po_front_end {
    int raw_w, raw_h;       /* captured frame */
    int undist_w, undist_h; /* undistoredWidth / Height */
    double K[9];            /* lens matrix at undist size (cameras.yaml K) */
    double dist[4];         /* k1 k2 p1 p2 */
    int rect[4];            /* crop x y w h */
    int out_w, out_h;       /* outPutWidth / Height = the stitcher's frame size */
} po_front_end;
/* cv::getOptimalNewCameraMatrix(K, dist, size, alpha = 1, size) (calib3d/src/calibration.cpp) */
void po_optimal_new_camera_matrix(const double K[9], const double dist[4], int w, int h, double newK[9]);
/* stitcher-frame coordinates -> raw-frame coordinates through the five inverse steps */
void po_front_end_map(const po_front_end* fe, const double newK[9], float xo, float yo, float* xr, float* yr);

/* ---- exposure: gain estimation (ocvstitcher.hpp:1031-1032, stitching_detailed.cpp:722-723) */
/* detail::GainCompensator::feed on sub-images given as pointer + stride; gains (n doubles) out; 0 = singular */
int po_gain_feed(int n, const int* corners, const int* sizes, const uint8_t* const* imgs, const size_t* istride,
                 const uint8_t* const* masks, const size_t* mstride, double* gains);
void po_gain_blocks_map_size(int w, int h, int bl_w, int bl_h, int wh[2]);
/* detail::BlocksGainCompensator::feed: images 8UC3 / masks 8U, both dense (stride = width); maps out */
int po_gain_blocks_feed(int n, const int* corners, const int* sizes, const uint8_t* const* imgs,
                        const uint8_t* const* masks, int bl_w, int bl_h, float** maps);
/* ocvStitcher::initSeam's feed of the compensator from stitcher-size frames (resize, seam-scale warps, feed) */
int po_estimate_gains(int n, int kind, int sw, int sh, const uint8_t* const* frames, const float* Ks, const float* Rs,
                      float wscale, int bl_w, int bl_h, int* seam_sizes, float** maps);

/* ---- whole per-frame path, ocvStitcher::process (ocvstitcher.hpp:1141-1216) */
typedef struct po_compose_args {
    int n;              /* num_images */
    int kind;           /* projector */
    int src_w, src_h;
    const uint8_t* const* frames;   /* n BGR8 interleaved, stride = src_w*3 */
    const float* K9s;   /* n*9 */
    const float* R9s;   /* n*9 */
    float scale;        /* warped_image_scale */
    const uint8_t* const* masks;    /* n, each warp_roi(i) sized (m_blenderMask) */
    int num_bands;      /* -1: Blender::NO */
    const float* const* gain_maps;  /* NULL, or n maps each warp_roi(i) sized (exposure apply) */
    int cut[4];         /* x,y,w,h inside dst_roi_final; w==0 -> full */
    const po_front_end* front; /* NULL, or n front ends: frames are then RAW frames of raw_w x raw_h */
} po_compose_args;
/* out must hold cut.w*cut.h*3 bytes (or the full pano).  Returns 0 on success. */
int po_compose(const po_compose_args* a, uint8_t* out, int out_wh[2]);
/* stage timings of the last po_compose on this thread, ms: warp, feed(pyramids+accumulate), blend */
void po_last_timings(double ms[3]);

/* ---- caller-side assembly of the two half panoramas (the step right after the path) */
/* cv::resize(src, dst, dsize) INTER_LINEAR, CV_8U (imgproc/src/resize.cpp: 11-bit fixed-point coefficients) */
void po_resize_linear_8u(const uint8_t* src, int sw, int sh, int cn, uint8_t* dst, int dw, int dh);
/* src/master.cpp:321-326: resize(up -> down.size()), vconcat, black 10-row bar; out is dw x 2*dh */
void po_stack_master(const uint8_t* up, int uw, int uh, const uint8_t* down, int dw, int dh, uint8_t* out);
/* src/panocamimpl.cpp:354-360: crop both to min width, min height - 2*finalcut (from row finalcut), vconcat,
 * black 4-row bar; out is min_w x 2*(min_h - 2*finalcut) */
void po_stack_finalcut(const uint8_t* up, int uw, int uh, const uint8_t* down, int dw, int dh, int finalcut, uint8_t* out);

void po_set_threads(int n);
int po_get_threads(void);

#ifdef __cplusplus
}
#endif
#endif
