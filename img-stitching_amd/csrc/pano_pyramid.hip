// pano_pyramid.hip - K2: cv::pyrDown on planar u8
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

// ------------------------------------------------------------------------------------------------
// K2: pyrDown (cv::pyrDown CV_16S semantics: 5x5 [1 4 6 4 1]^2, REFLECT_101, (v+128)>>8) on planar u8.
// One thread = 4 x 2 output pixels of one plane: 7 input rows x one 16-byte load, the horizontal 5-tap as
// v_dot4_u32_u8 on byte windows, the vertical pass in registers.  grid.z = camera * 3 + plane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pyr_down_kernel(PyrParams P, unsigned cam_bits, int l) {
    const int ci = blockIdx.z / 3, pl = blockIdx.z - ci * 3;
    if (!((cam_bits >> ci) & 1u)) return;
    const PyrCam& c = P.cam[ci];
    const int sw = c.w0 >> l, sh = c.h0 >> l;
    const int dw = sw >> 1, dh = sh >> 1;
    // Outputs of level l + 1 that nothing downstream reads are not produced (their inputs may not exist either): the
    // grid is laid over the live rect, so that whole waves - not lanes - fall off its far side.
    const int lx0 = c.live[l + 1][0], ly0 = c.live[l + 1][1], lx1 = c.live[l + 1][2], ly1 = c.live[l + 1][3];
    const int t = (lx0 >> 2) + blockIdx.x * 64 + threadIdx.x;  // group of 4 output columns
    // a wave is one threadIdx.y: tell the compiler, and the row indices, the REFLECT_101 of the seven source rows and
    // their addresses are scalar work (a quarter of this kernel's vector instructions otherwise)
    const int y0 = ((ly0 >> 1) + blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * 2;  // pair of output rows
    if (y0 >= dh || y0 > ly1) return;
    if (t * 4 >= dw || t * 4 > lx1) return;
    if (t * 4 >= c.gap[l + 1][0] && t * 4 + 3 <= c.gap[l + 1][1]) return;  // the dead middle of a +-pi straddler's tile
    const uint8_t* __restrict__ src = c.lvl[l] + (size_t)pl * c.plane[l];
    const int sp = c.pitch[l];
    // Every lane does seven 16-byte loads (columns 8t-4 .. 8t+11; lane 0 loads columns 0..15 and shifts).
    // REFLECT_101 at the two row ends touches at most three bytes, patched in registers:
    //   left  (t == 0): columns -2, -1 are columns 2, 1
    //   right (8t+8 == sw, the last group): column sw is column sw-2
    int acc0[4] = {128, 128, 128, 128}, acc1[4] = {128, 128, 128, 128};  // the rounding term of (v + 128) >> 8
    const int off = t == 0 ? 0 : 8 * t - 4;
    uint4 q[7];
#pragma unroll
    for (int r = 0; r < 7; r++)
        q[r] = *reinterpret_cast<const uint4*>(src + ((unsigned)(reflect101_idx(2 * y0 - 2 + r, sh) * sp) + (unsigned)off));  // scalar row + lane offset
    if (t == 0) {
#pragma unroll
        for (int r = 0; r < 7; r++) {
            const unsigned a = q[r].x;
            q[r].w = q[r].z;
            q[r].z = q[r].y;
            q[r].y = a;
            // byte2 = column -2 = column 2 (column 0 when the row has only two), byte3 = column -1 = column 1
            q[r].x = (sw > 2 ? (a & 0x00ff0000u) : ((a & 0xffu) << 16)) | ((a & 0x0000ff00u) << 16);
        }
    }
    const int ksw = sw - (8 * t - 4);  // byte position of column sw in the 16-byte window
    if (ksw <= 12) {
        // ksw is even (sw and 8t-4 are even) and >= 6: the source byte ksw-2 sits in the same or the previous dword
#pragma unroll
        for (int r = 0; r < 7; r++) {
            unsigned d[4] = {q[r].x, q[r].y, q[r].z, q[r].w};
            const int ks = ksw - 2;
            unsigned v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((ks >> 2) == j) v = (d[j] >> (8 * (ks & 3))) & 0xffu;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if ((ksw >> 2) == j) d[j] = (d[j] & ~(0xffu << (8 * (ksw & 3)))) | (v << (8 * (ksw & 3)));
            q[r] = make_uint4(d[0], d[1], d[2], d[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 7; r++) {
        int h[4];
        pyr_down_hrow(q[r], h);
        const int wa = r == 0 ? 1 : (r == 1 ? 4 : (r == 2 ? 6 : (r == 3 ? 4 : (r == 4 ? 1 : 0))));
        const int wb = r == 2 ? 1 : (r == 3 ? 4 : (r == 4 ? 6 : (r == 5 ? 4 : (r == 6 ? 1 : 0))));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc0[j] += h[j] * wa;
            acc1[j] += h[j] * wb;
        }
    }
    uint8_t* d = c.lvl[l + 1] + (size_t)pl * c.plane[l + 1] + (size_t)y0 * c.pitch[l + 1] + 4 * t;
    unsigned p0 = 0, p1 = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        // no saturate_cast: the taps sum to 256, so (256 * 255 + 128) >> 8 = 255 is the largest value there is
        p0 |= (unsigned)(acc0[j] >> 8) << (8 * j);
        p1 |= (unsigned)(acc1[j] >> 8) << (8 * j);
    }
    *reinterpret_cast<unsigned*>(d) = p0;  // rows are padded to 16 bytes
    if (y0 + 1 < dh) *reinterpret_cast<unsigned*>(d + c.pitch[l + 1]) = p1;
}

void launch_pyr_down(const PyrParams& p, unsigned cam_bits, int l, hipStream_t s) {
    int mw = 0, mh = 0;
    for (int i = 0; i < p.ncam; i++)
        if ((cam_bits >> i) & 1u) {
            mw = max(mw, p.cam[i].w0 >> (l + 1));
            mh = max(mh, p.cam[i].h0 >> (l + 1));
        }
    if (mw == 0 || mh == 0) return;
    dim3 block(64, 4, 1), grid((mw + 255) / 256, (mh + 7) / 8, p.ncam * 3);
    hipLaunchKernelGGL(pyr_down_kernel, grid, block, 0, s, p, cam_bits, l);
}

}  // namespace pano
