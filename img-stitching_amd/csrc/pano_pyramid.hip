// pano_pyramid.hip - K2: cv::pyrDown on planar u8
// Device helpers: pano_dev.hpp; launch interface: pano_kernels.hpp.  Compile with -ffp-contract=off.

#include "pano_dev.hpp"

namespace pano {

// ------------------------------------------------------------------------------------------------
// K2: pyrDown (cv::pyrDown CV_16S semantics: 5x5 [1 4 6 4 1]^2, REFLECT_101, (v+128)>>8) on planar u8.
// One thread = 4 x R output pixels of one plane: 2R + 3 input rows x one 16-byte load, the horizontal 5-tap as
// v_dot4_u32_u8 on byte windows, the vertical pass in registers.  grid.z = camera * 3 + plane.
// R = 4 (eleven loads for sixteen outputs, 68 VGPRs, 7 waves / SIMD): against R = 2 a lane fetches 2.75 instead of 3.5 rows
// per output row and does as many horizontal passes, and half as many waves pay the scalar prologue - the pyramid stage of
// config 2 33.5 -> 31.7 us one frame at a time, +1.5 % panoramas/s with four in flight; R = 5, 6, 8 (92 / 116 VGPRs) measure the same.
// ------------------------------------------------------------------------------------------------
#ifndef PANO_PYR_ROWS
#define PANO_PYR_ROWS 4
#endif
#ifndef PANO_PYR_ROWS_UP
#define PANO_PYR_ROWS_UP 4
#endif
constexpr int kPyrRows0 = PANO_PYR_ROWS, kPyrRowsUp = PANO_PYR_ROWS_UP;  // output rows per lane: level 0 -> 1, the levels above
// The deal (like K1's, pano_kernels.hpp WarpDeal): the LIVE workgroups of every camera of the launch - camera after camera, plane
// after plane, row after row - form one list and XCD k (= blockIdx.x: grid.x is 8) takes the k-th eighth of it: no workgroup is
// launched only to find its rows dead, every XCD has the same number, and a plane's neighbouring workgroups - which share three rows
// and a 128-byte line each - sit in one L2.  The prefix sums (e0 .. e7: workgroups of cameras 0 .. c) and the share per XCD lead the
// arguments (preloaded into SGPRs at wave launch, see the Makefile: no memory round trip); the level rides in per's top bits.
// Prologue: the camera's fields of the two levels and its deal entry are requested TOGETHER - seven scalar loads, one wait.  Read
// field by field behind the early exits they were seven dependent round trips per wave (about 1 us of a wave that lives 2 - 3).
constexpr unsigned kPyrParamsAt = 40;  // offset of P in the kernarg segment: nine dwords in front of it, then alignof(PyrParams)
template <int R>
__global__ __launch_bounds__(256) void pyr_down_kernel(unsigned e0, unsigned e1, unsigned e2, unsigned e3, unsigned e4, unsigned e5, unsigned e6,
                                                       unsigned e7, unsigned per_l, PyrParams P) {
    static_assert(alignof(PyrParams) == 8 && kCams == 8, "kPyrParamsAt, eight prefix sums");
    constexpr int NR = 2 * R + 3;  // source rows of R output rows
    const unsigned per = per_l & 0x0fffffffu;
    const int l = (int)(per_l >> 28);
    const unsigned lin = blockIdx.x * per + blockIdx.y;
    if (lin >= e7) return;  // past the end of the list (the whole workgroup leaves)
    const unsigned ends[kCams] = {e0, e1, e2, e3, e4, e5, e6, e7};
    unsigned ci = 0, dstart = 0;
#pragma unroll
    for (int c = 0; c < kCams - 1; c++) {
        if (lin >= ends[c]) {
            ci = c + 1;
            dstart = ends[c];
        }
    }
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
    u32x4 cdeal;
    {
        const char __attribute__((address_space(4)))* a_deal = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
            (kPyrParamsAt + (unsigned)offsetof(PyrParams, deal) + ci * 16u);
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(cdeal) : "s"(a_deal) : "memory");
    }
    i32x2 cwh, cgap, cpitch, cplane;
    i32x4 clive;
    u64x2 clvl;
    {
        const char __attribute__((address_space(4)))* cb =
            (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + (kPyrParamsAt + (unsigned)ci * sizeof(PyrCam));
        const char __attribute__((address_space(4)))* a_live = cb + (offsetof(PyrCam, live) + (unsigned)(l + 1) * 16u);
        const char __attribute__((address_space(4)))* a_gap = cb + (offsetof(PyrCam, gap) + (unsigned)(l + 1) * 8u);
        const char __attribute__((address_space(4)))* a_lvl = cb + (offsetof(PyrCam, lvl) + (unsigned)l * 8u);
        const char __attribute__((address_space(4)))* a_pitch = cb + (offsetof(PyrCam, pitch) + (unsigned)l * 4u);
        const char __attribute__((address_space(4)))* a_plane = cb + (offsetof(PyrCam, plane) + (unsigned)l * 4u);
        const char __attribute__((address_space(4)))* a_wh = cb + offsetof(PyrCam, w0);
        static_assert(offsetof(PyrCam, h0) == offsetof(PyrCam, w0) + 4, "w0, h0 adjacent");
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(cwh) : "s"(a_wh) : "memory");
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(clive) : "s"(a_live) : "memory");
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(cgap) : "s"(a_gap) : "memory");
        asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=&s"(clvl) : "s"(a_lvl) : "memory");
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(cpitch) : "s"(a_pitch) : "memory");
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(cplane) : "s"(a_plane) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(cwh), "+s"(clive), "+s"(cgap), "+s"(clvl), "+s"(cpitch), "+s"(cplane), "+s"(cdeal) : : "memory");
    }
    // position in the camera's list -> plane, workgroup row, workgroup column (divisions by multiplication)
    const unsigned dl = lin - dstart;
    const unsigned pl = cdeal.w ? __umulhi(dl, cdeal.w) : dl;
    const unsigned dr = dl - pl * cdeal.z;
    const unsigned by = cdeal.y ? __umulhi(dr, cdeal.y) : dr, bx = dr - by * cdeal.x;
    const int sw = cwh.x >> l, sh = cwh.y >> l;
    const int dw = sw >> 1, dh = sh >> 1;
    // Outputs of level l + 1 that nothing downstream reads are not produced (their inputs may not exist either): the
    // grid is laid over the live rect, so that whole waves - not lanes - fall off its far side.
    const int lx0 = clive.x, ly0 = clive.y, lx1 = clive.z, ly1 = clive.w;
    const int t = (lx0 >> 2) + (int)bx * 64 + threadIdx.x;  // group of 4 output columns
    // a wave is one threadIdx.y: tell the compiler, and the row indices, the REFLECT_101 of the source rows and
    // their addresses are scalar work (a quarter of this kernel's vector instructions otherwise)
    const int y0 = ((int)((unsigned)ly0 / R) + (int)by * 4 + __builtin_amdgcn_readfirstlane(threadIdx.y)) * R;  // first of R output rows
    if (y0 >= dh || y0 > ly1) return;
    if (t * 4 >= dw || t * 4 > lx1) return;
    if (t * 4 >= cgap.x && t * 4 + 3 <= cgap.y) return;  // the dead middle of a +-pi straddler's tile
    const uint8_t* __restrict__ src = (const uint8_t*)clvl.x + (size_t)pl * cplane.x;
    const int sp = cpitch.x;
    // Every lane does NR 16-byte loads (columns 8t-4 .. 8t+11; lane 0 loads columns 0..15 and shifts).
    // REFLECT_101 at the two row ends touches at most three bytes, patched in registers:
    //   left  (t == 0): columns -2, -1 are columns 2, 1
    //   right (8t+8 == sw, the last group): column sw is column sw-2
    const int off = t == 0 ? 0 : 8 * t - 4;
    uint4 q[NR];
#pragma unroll
    for (int r = 0; r < NR; r++)
    {   // one 16-byte load at a 4-byte aligned address (left to itself the compiler made it an 8- and a 12-byte load that overlap)
        typedef unsigned u32x4a __attribute__((ext_vector_type(4), aligned(4)));
        const u32x4a v = *reinterpret_cast<const u32x4a*>(src + ((unsigned)(reflect101_idx(2 * y0 - 2 + r, sh) * sp) + (unsigned)off));  // scalar row + lane offset
        q[r] = make_uint4(v.x, v.y, v.z, v.w);
    }
    if (t == 0) {
#pragma unroll
        for (int r = 0; r < NR; r++) {
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
        for (int r = 0; r < NR; r++) {
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
    int acc[R][4];
#pragma unroll
    for (int o = 0; o < R; o++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[o][j] = 128;  // the rounding term of (v + 128) >> 8
#pragma unroll
    for (int r = 0; r < NR; r++) {
        int h[4];
        pyr_down_hrow(q[r], h);
#pragma unroll
        for (int o = 0; o < R; o++) {
            const int k = r - 2 * o;  // tap of output row o that source row r is
            if (k < 0 || k > 4) continue;
            const int wk = k == 0 || k == 4 ? 1 : (k == 2 ? 6 : 4);
#pragma unroll
            for (int j = 0; j < 4; j++) acc[o][j] += h[j] * wk;
        }
    }
    uint8_t* d = (uint8_t*)clvl.y + (size_t)pl * cplane.y + (size_t)y0 * cpitch.y + 4 * t;
#pragma unroll
    for (int o = 0; o < R; o++) {
        unsigned px = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) px |= (unsigned)(acc[o][j] >> 8) << (8 * j);  // no saturate_cast: the taps sum to 256, so 255 is the largest value there is
        if (o == 0 || y0 + o < dh) *reinterpret_cast<unsigned*>(d + (size_t)o * cpitch.y) = px;  // rows are padded to 16 bytes
    }
}

// ------------------------------------------------------------------------------------------------
// K2 tail: the pyrDown chain level b -> b+1 -> ... -> t (t - b <= 4) of a camera plane in ONE launch.
// One at a time the small levels are launches of 4 - 5 us with almost no work (config 2: 216 x 124 pixels and fewer per
// plane above level 2): every launch pays its ramp - dispatch, cold instruction cache, a round trip of dependent loads, the
// drain of its stores - and the chain pays it per level.  Here a workgroup owns a TS x TS tile of level b+1 (and the tiles
// half, a quarter, an eighth that size of the levels above that lie over it), stages the pixels of level b that its chain
// depends on in LDS once (the footprint grows by 2 pixels per side and level, h_l = 2 h_{l+1} + 2: at most 30 per side), and
// walks up level by level through LDS with the halo recomputed; every level's owned pixels are written on the way.
// Same arithmetic as pyr_down_kernel bit for bit: 4 x 2 outputs per lane and step from seven 16-byte windows, pyr_down_hrow's
// chained v_dot4.  The LDS boxes hold REFLECT_101 pads (rows -2, -1, h; columns -2, -1, w: written by whoever writes the cell
// they mirror, in workgroups whose boxes touch a border), so the windows are read with no border logic: three ds_read_b64
// per row, of which the middle 16 bytes are the window.
// What is produced follows the live rects like pyr_down_kernel: per level the owned tile is cut to the live rect, what
// is computed is the bounding box of that and of the footprint of the level above ("need"), nothing else.
// Needs every source level (b .. t-1) to be at least 4 x 4 (the pads mirror rows / columns 1 and 2): pyr_tail_ok.
// ------------------------------------------------------------------------------------------------
constexpr int kTailMaxJ = 4;
constexpr int kTailSlack = 16;  // bytes in front of and behind every LDS row: the column pads, and the part of a 24-byte
                                // read that lies outside the stored columns (it feeds outputs nobody needs)
// worst-case LDS box of level b + j for a TS x TS tile of level b + 1 and a chain of at most MJ levels (pitch x rows incl. 2
// pad rows on top and 3 below): halo 2^(MJ - j + 1) - 2 pixels per side; columns stored from a multiple of 16 (level b) / 8
__host__ __device__ constexpr int tail_halo(int MJ, int j) { return j >= MJ ? 0 : (2 << (MJ - j)) - 2; }
__host__ __device__ constexpr int tail_tile(int TS, int j) { return j == 0 ? 2 * TS : TS >> (j - 1); }
__host__ __device__ constexpr int tail_pitch(int TS, int MJ, int j) {
    return (tail_tile(TS, j) < 8 ? 8 : tail_tile(TS, j)) + 2 * ((tail_halo(MJ, j) + (j == 0 ? 15 : 7)) & ~(j == 0 ? 15 : 7)) + 2 * kTailSlack;
}
__host__ __device__ constexpr int tail_cap(int TS, int MJ, int j) {
    return j > MJ ? 0 : tail_pitch(TS, MJ, j) * (tail_tile(TS, j) + 2 * tail_halo(MJ, j) + 5);
}
__host__ __device__ constexpr int tail_lds_bytes(int TS, int MJ) {
    return tail_cap(TS, MJ, 0) + tail_cap(TS, MJ, 1) + tail_cap(TS, MJ, 2) + tail_cap(TS, MJ, 3) + tail_cap(TS, MJ, 4);
}
static_assert(tail_lds_bytes(kPyrTailTile, 4) <= 64 * 1024, "static LDS");
struct TailBox {
    int x0, y0, x1, y1;  // inclusive; x1 < x0: empty
};
__device__ __forceinline__ bool tail_empty(const TailBox& b) { return b.x1 < b.x0 || b.y1 < b.y0; }

// Stores N dwords d[] = columns x .. x + 4N - 1 of row y of a w x h level into its LDS box (row 0 of the box = level row
// by0 - 2, byte kTailSlack of a box row = level column bx0); EDGE: also into the REFLECT_101 pads that mirror them
template <int N, bool EDGE>
__device__ __forceinline__ void tail_put(uint8_t* box, int pitch, int bx0, int by0, int w, int h, int x, int y, const unsigned (&d)[N]) {
    const int rows[3] = {y, (y == 1 || y == 2) ? -y : INT_MIN, y == h - 2 ? h : INT_MIN};
#pragma unroll
    for (int k = 0; k < (EDGE ? 3 : 1); k++) {
        if (rows[k] == INT_MIN) continue;
        uint8_t* r = box + (rows[k] - by0 + 2) * pitch + kTailSlack + (x - bx0);
#pragma unroll
        for (int i = 0; i < N; i++) reinterpret_cast<unsigned*>(r)[i] = d[i];
        if (!EDGE) continue;
        if (x == 0) {  // columns -1, -2 = columns 1, 2
            r[-1] = (uint8_t)(d[0] >> 8);
            r[-2] = (uint8_t)(d[0] >> 16);
        }
        const int m = w - 2 - x;  // column w = column w - 2
        if (m >= 0 && m < 4 * N) {
            unsigned v = 0;
#pragma unroll
            for (int i = 0; i < N; i++)
                if ((m >> 2) == i) v = d[i];
            r[m + 2] = (uint8_t)(v >> (8 * (m & 3)));
        }
    }
}

template <int TS, int THREADS, int MJ>
__global__ __launch_bounds__(THREADS) void pyr_tail_kernel(unsigned cam_bits, int b, int t, PyrParams P) {
    static_assert(MJ >= 1 && MJ <= kTailMaxJ && tail_cap(TS, MJ, 0) % 8 == 0 && tail_cap(TS, MJ, 1) % 8 == 0 && tail_cap(TS, MJ, 2) % 8 == 0 &&
                      tail_cap(TS, MJ, 3) % 8 == 0, "LDS boxes 8-byte aligned");
    __shared__ __attribute__((aligned(16))) uint8_t lds[tail_lds_bytes(TS, MJ)];
    const int ci = blockIdx.z / 3, pl = blockIdx.z - ci * 3;
    if (!((cam_bits >> ci) & 1u)) return;
    // The camera's fields of levels b .. b + 4, requested TOGETHER (cam_bits, b, t are preloaded): read where they are used -
    // level after level, behind the block-uniform tests - they were fourteen dependent scalar round trips in front of the first
    // row load, in a launch that is a single round of waves.  b <= kLevels - 5, so the reads stay inside PyrCam.
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    typedef int i32x8 __attribute__((ext_vector_type(8)));
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned long long u64x8 __attribute__((ext_vector_type(8)));
    struct {
        i32x2 wh;      // w0, h0
        i32x16 live;   // live[b + 1 .. b + 4][4]
        i32x8 gap;     // gap[b + 1 .. b + 4][2]
        u64x8 lvl;     // lvl[b .. b + 7] (the first five are used)
        i32x8 pitch, plane;  // [b .. b + 7]
    } c;
    {
        static_assert(alignof(PyrParams) == 8 && kLevels >= 9, "P sits 16 bytes into the kernarg segment; b + 4 < kLevels");
        const char __attribute__((address_space(4)))* cb =
            (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + (16u + (unsigned)ci * sizeof(PyrCam));
        const char __attribute__((address_space(4)))* a_wh = cb + offsetof(PyrCam, w0);
        const char __attribute__((address_space(4)))* a_live = cb + (offsetof(PyrCam, live) + (unsigned)(b + 1) * 16u);
        const char __attribute__((address_space(4)))* a_gap = cb + (offsetof(PyrCam, gap) + (unsigned)(b + 1) * 8u);
        const char __attribute__((address_space(4)))* a_lvl = cb + (offsetof(PyrCam, lvl) + (unsigned)b * 8u);
        const char __attribute__((address_space(4)))* a_pitch = cb + (offsetof(PyrCam, pitch) + (unsigned)b * 4u);
        const char __attribute__((address_space(4)))* a_plane = cb + (offsetof(PyrCam, plane) + (unsigned)b * 4u);
        static_assert(offsetof(PyrCam, lvl) + 4 * 8 + 64 <= sizeof(PyrCam) && offsetof(PyrCam, pitch) + 4 * 4 + 32 <= sizeof(PyrCam) &&
                          offsetof(PyrCam, plane) + 4 * 4 + 32 <= sizeof(PyrCam) && offsetof(PyrCam, live) + 5 * 16 + 64 <= sizeof(PyrCam) &&
                          offsetof(PyrCam, gap) + 5 * 8 + 32 <= sizeof(PyrCam), "the batched reads stay inside PyrCam for b <= 4");
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(c.wh) : "s"(a_wh) : "memory");
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(c.live) : "s"(a_live) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(c.gap) : "s"(a_gap) : "memory");
        asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(c.lvl) : "s"(a_lvl) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(c.pitch) : "s"(a_pitch) : "memory");
        asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=&s"(c.plane) : "s"(a_plane) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(c.wh), "+s"(c.live), "+s"(c.gap), "+s"(c.lvl), "+s"(c.pitch), "+s"(c.plane) : : "memory");
    }
    const int J = t - b;  // <= MJ
    const int tid = threadIdx.y * 64 + threadIdx.x;
    // what this workgroup owns and needs of every level, top down (block-uniform: scalar registers)
    TailBox own[kTailMaxJ + 1], need[kTailMaxJ + 1];
    TailBox nx = {0, 0, -1, -1};
#pragma unroll
    for (int j = kTailMaxJ; j >= 1; j--) {
        own[j] = need[j] = TailBox{0, 0, -1, -1};
        if (j > J || j > MJ) continue;
        const int l = b + j;
        const int Tl = TS >> (j - 1);
        const int w = c.wh.x >> l, h = c.wh.y >> l;
        TailBox o;
        o.x0 = max((int)blockIdx.x * Tl, c.live[4 * (j - 1) + 0]);
        o.x1 = min(min((int)blockIdx.x * Tl + Tl - 1, w - 1), c.live[4 * (j - 1) + 2]);
        o.y0 = max((int)blockIdx.y * Tl, c.live[4 * (j - 1) + 1]);
        o.y1 = min(min((int)blockIdx.y * Tl + Tl - 1, h - 1), c.live[4 * (j - 1) + 3]);
        if (o.x0 >= c.gap[2 * (j - 1)] && o.x1 <= c.gap[2 * (j - 1) + 1]) o.x1 = o.x0 - 1;  // inside the dead middle of a +-pi straddler's tile
        if (tail_empty(o)) o = TailBox{0, 0, -1, -1};
        own[j] = o;
        TailBox n = o;
        if (!tail_empty(nx)) {  // the pyrDown footprint of what the level above needs
            const TailBox f = {max(2 * nx.x0 - 2, 0), max(2 * nx.y0 - 2, 0), min(2 * nx.x1 + 2, w - 1), min(2 * nx.y1 + 2, h - 1)};
            n = tail_empty(o) ? f : TailBox{min(o.x0, f.x0), min(o.y0, f.y0), max(o.x1, f.x1), max(o.y1, f.y1)};
        }
        need[j] = n;
        nx = n;
    }
    if (tail_empty(nx)) return;  // nothing of this tile is live at any level (the whole workgroup leaves)
    {
        const int w = c.wh.x >> b, h = c.wh.y >> b;
        own[0] = TailBox{0, 0, -1, -1};
        need[0] = TailBox{max(2 * nx.x0 - 2, 0), max(2 * nx.y0 - 2, 0), min(2 * nx.x1 + 2, w - 1), min(2 * nx.y1 + 2, h - 1)};
    }
    // LDS boxes: columns from sx0 (16-byte aligned at level b, else 8-byte), rows from sy0 - 2; edge: the box holds cells
    // that a REFLECT_101 pad mirrors (rows 1, 2, h - 2; columns 1, 2, w - 2)
    int sx0[kTailMaxJ + 1], sy0[kTailMaxJ + 1], pitch[kTailMaxJ + 1], off[kTailMaxJ + 1];
    bool edge[kTailMaxJ + 1];
    {
        int o = 0;
#pragma unroll
        for (int j = 0; j <= kTailMaxJ; j++) {
            const int ax = j == 0 ? 15 : 7;
            const int w = c.wh.x >> (b + j), h = c.wh.y >> (b + j);
            sx0[j] = need[j].x0 & ~ax;
            sy0[j] = need[j].y0;
            pitch[j] = tail_empty(need[j]) ? 0 : ((need[j].x1 | ax) - sx0[j] + 1) + 2 * kTailSlack;
            edge[j] = sx0[j] == 0 || (need[j].x1 | ax) >= w - 2 || need[j].y0 <= 2 || need[j].y1 >= h - 2;
            off[j] = o;
            o += tail_cap(TS, MJ, j);
        }
    }
    // level b: global -> LDS, 16-byte chunks; lanes = 16 chunk columns x THREADS / 16 rows, every load in flight before the
    // first store
    {
        const int w = c.wh.x >> b, h = c.wh.y >> b;
        const int nrows = need[0].y1 - need[0].y0 + 1;  // <= 2 TS + 2 halo
        const uint8_t* g = (const uint8_t*)c.lvl[0] + (size_t)pl * c.plane[0] + (unsigned)(sy0[0] * c.pitch[0] + sx0[0]);
        const int cpr = (pitch[0] - 2 * kTailSlack) >> 4;
        static_assert((tail_pitch(TS, MJ, 0) - 2 * kTailSlack) / 16 <= 16, "16 chunk columns");
        const int cx = tid & 15, cy = tid >> 4;
        if (cx < cpr) {
            constexpr int kRowsPerPass = THREADS / 16;
            constexpr int kBatch = (2 * TS + 2 * tail_halo(MJ, 0) + kRowsPerPass - 1) / kRowsPerPass;
            uint4 d[kBatch];
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int r = min(cy + kRowsPerPass * k, nrows - 1);
                d[k] = *reinterpret_cast<const uint4*>(g + (unsigned)(r * c.pitch[0] + 16 * cx));
            }
#pragma unroll
            for (int k = 0; k < kBatch; k++) {
                const int r = cy + kRowsPerPass * k;
                if (r < nrows) {
                    const unsigned dd[4] = {d[k].x, d[k].y, d[k].z, d[k].w};
                    if (edge[0]) tail_put<4, true>(lds + off[0], pitch[0], sx0[0], sy0[0], w, h, sx0[0] + 16 * cx, sy0[0] + r, dd);
                    else tail_put<4, false>(lds + off[0], pitch[0], sx0[0], sy0[0], w, h, sx0[0] + 16 * cx, sy0[0] + r, dd);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 1; j <= MJ; j++) {
        if (j > J || tail_empty(need[j])) continue;  // block-uniform
        const int l = b + j;
        const int dw = c.wh.x >> l, dh = c.wh.y >> l;
        const uint8_t* S = lds + off[j - 1];
        const int ps = pitch[j - 1];
        // items = column groups need.x0 >> 2 .. need.x1 >> 2 (<= 24) x row pairs need.y0 >> 1 .. need.y1 >> 1 (<= 47), dealt
        // to the lanes in row-major order: the waves beyond the last item leave at once
        const int g0 = need[j].x0 >> 2, ngx = (need[j].x1 >> 2) - g0 + 1;
        const int r0 = need[j].y0 >> 1, nry = (need[j].y1 >> 1) - r0 + 1;
        const int nitems = ngx * nry;
        const unsigned rcp = (65536u + (unsigned)ngx - 1u) / (unsigned)ngx;  // item / ngx == item * rcp >> 16 for item < 2730
        uint8_t* const gl = (uint8_t*)c.lvl[j] + (size_t)pl * c.plane[j];
        const int gp = c.pitch[j];
        for (int item = tid; item < nitems; item += THREADS) {
            const int rp = (int)(((unsigned)item * rcp) >> 16);
            const int tg = g0 + item - rp * ngx;  // group of four output columns 4 tg .. 4 tg + 3
            const int y0 = 2 * (r0 + rp);
            // source bytes 8 tg - 8 .. 8 tg + 15 (8-byte aligned in LDS); the window is columns 8 tg - 4 .. 8 tg + 11
            const uint8_t* Sr = S + (2 * y0 - sy0[j - 1]) * ps + kTailSlack + (8 * tg - 8 - sx0[j - 1]);
            int acc0[4] = {128, 128, 128, 128}, acc1[4] = {128, 128, 128, 128};
#pragma unroll
            for (int r = 0; r < 7; r++) {
                const uint2* r64 = reinterpret_cast<const uint2*>(Sr + r * ps);
                const uint2 A = r64[0], B = r64[1], C = r64[2];
                int h[4];
                pyr_down_hrow(make_uint4(A.y, B.x, B.y, C.x), h);
                const int wa = r == 0 ? 1 : (r == 1 ? 4 : (r == 2 ? 6 : (r == 3 ? 4 : (r == 4 ? 1 : 0))));
                const int wb = r == 2 ? 1 : (r == 3 ? 4 : (r == 4 ? 6 : (r == 5 ? 4 : (r == 6 ? 1 : 0))));
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    acc0[k] += h[k] * wa;
                    acc1[k] += h[k] * wb;
                }
            }
            unsigned p0 = 0, p1 = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                p0 |= (unsigned)(acc0[k] >> 8) << (8 * k);
                p1 |= (unsigned)(acc1[k] >> 8) << (8 * k);
            }
            if (j < J) {  // this level is the source of the next
                const unsigned d0[1] = {p0}, d1[1] = {p1};
                if (edge[j]) {
                    tail_put<1, true>(lds + off[j], pitch[j], sx0[j], sy0[j], dw, dh, 4 * tg, y0, d0);
                    tail_put<1, true>(lds + off[j], pitch[j], sx0[j], sy0[j], dw, dh, 4 * tg, y0 + 1, d1);
                } else {
                    tail_put<1, false>(lds + off[j], pitch[j], sx0[j], sy0[j], dw, dh, 4 * tg, y0, d0);
                    tail_put<1, false>(lds + off[j], pitch[j], sx0[j], sy0[j], dw, dh, 4 * tg, y0 + 1, d1);
                }
            }
            if (4 * tg + 3 >= own[j].x0 && 4 * tg <= own[j].x1) {
                if (y0 >= own[j].y0 && y0 <= own[j].y1) *reinterpret_cast<unsigned*>(gl + (unsigned)(y0 * gp + 4 * tg)) = p0;
                if (y0 + 1 >= own[j].y0 && y0 + 1 <= own[j].y1) *reinterpret_cast<unsigned*>(gl + (unsigned)((y0 + 1) * gp + 4 * tg)) = p1;
            }
        }
        if (j < J) __syncthreads();
    }
}

// the tail needs source levels of at least 4 x 4
bool pyr_tail_ok(const PyrParams& p, unsigned cam_bits, int t) {
    for (int i = 0; i < p.ncam; i++)
        if (((cam_bits >> i) & 1u) && ((p.cam[i].w0 >> (t - 1)) < 4 || (p.cam[i].h0 >> (t - 1)) < 4)) return false;
    return true;
}

// one workgroup of 256 lanes owns a 32 x 32 tile of level b + 1 (64 x 64 tiles on 512 lanes measured slower)
void launch_pyr_tail(const PyrParams& p, unsigned cam_bits, int b, int t, hipStream_t s) {
    int mw = 0, mh = 0;
    for (int i = 0; i < p.ncam; i++)
        if ((cam_bits >> i) & 1u) {
            mw = max(mw, p.cam[i].w0 >> t);
            mh = max(mh, p.cam[i].h0 >> t);
        }
    const int J = t - b;
    if (mw == 0 || mh == 0 || J < 1 || J > kTailMaxJ) return;
    const int T = kPyrTailTile >> (J - 1);  // tile of the top level
    if (T < 1) return;
    dim3 grid((mw + T - 1) / T, (mh + T - 1) / T, p.ncam * 3);
#define PANO_TAIL(MJ_) hipLaunchKernelGGL((pyr_tail_kernel<kPyrTailTile, 256, MJ_>), grid, dim3(64, 4, 1), 0, s, cam_bits, b, t, p)
    if (J <= 2) PANO_TAIL(2);
    else if (J == 3) PANO_TAIL(3);
    else PANO_TAIL(4);
#undef PANO_TAIL
}

void launch_pyr_down(const PyrParams& p, unsigned cam_bits, int l, hipStream_t s) {
    const int R = l == 0 ? kPyrRows0 : kPyrRowsUp;
    PyrParams q = p;
    unsigned ends[kCams], total = 0;
    auto magic = [](unsigned d) { return d > 1 ? (unsigned)((1ull << 32) / d + 1ull) : 0u; };
    for (int i = 0; i < kCams; i++) {
        unsigned cols = 0, rows = 0;
        if (i < p.ncam && ((cam_bits >> i) & 1u)) {
            const PyrCam& c = p.cam[i];
            const int dw = (c.w0 >> l) >> 1, dh = (c.h0 >> l) >> 1;
            const int lx0 = c.live[l + 1][0], ly0 = c.live[l + 1][1], lx1 = min(c.live[l + 1][2], dw - 1), ly1 = min(c.live[l + 1][3], dh - 1);
            if (lx0 >= 0 && ly0 >= 0 && lx1 >= lx0 && ly1 >= ly0) {
                // the kernel's own tests: a workgroup's first lane is column group (lx0 >> 2) + 64 bx, its first wave's rows start at
                // (ly0 / R + 4 by) * R
                cols = (unsigned)(((lx1 >> 2) - (lx0 >> 2)) / 64 + 1);
                rows = (unsigned)((ly1 / R - ly0 / R) / 4 + 1);
            }
        }
        q.deal[i][0] = max(cols, 1u);
        q.deal[i][1] = magic(q.deal[i][0]);
        q.deal[i][2] = max(cols * rows, 1u);
        q.deal[i][3] = magic(q.deal[i][2]);
        total += 3u * cols * rows;
        ends[i] = total;
    }
    if (total == 0) return;
    const unsigned per = (total + 7u) / 8u;
    // (per is grid.y: a launch of more than 65535 x 8 workgroups - half a gigapixel of level l + 1 - fails in hipLaunchKernel and
    // surfaces as the entry's error, like K1's)
    const unsigned per_l = per | (unsigned)l << 28;
    const dim3 block(64, 4, 1), grid(8, per, 1);
    if (l == 0)
        hipLaunchKernelGGL(pyr_down_kernel<kPyrRows0>, grid, block, 0, s, ends[0], ends[1], ends[2], ends[3], ends[4], ends[5], ends[6], ends[7], per_l, q);
    else
        hipLaunchKernelGGL(pyr_down_kernel<kPyrRowsUp>, grid, block, 0, s, ends[0], ends[1], ends[2], ends[3], ends[4], ends[5], ends[6], ends[7], per_l, q);
}

}  // namespace pano
