// post_kernels.hip -- gfx950 kernels for the reference's post-processing chain (test.m:45-50).
// The MATLAB originals are sequential (raster scans, a FIFO flood fill, first-come writes); each
// kernel computes the same result in an order-free form and says why that is the same thing.
#include "post_kernels.h"
#include "fsgm_device.h"

namespace fsgm {

// =============================================================================================
// speckle_filter.m:1-103.  The flood fill (:43-92) joins a pixel to a neighbour when both are valid
// and |a-b| < maxDiff -- a symmetric relation, so its regions are the connected components of that
// graph whatever the seed order; a region is dropped when it has fewer than maxSpeckleSize pixels
// (:94-97, and :27-30 for its later pixels).  Components by lock-free union-find: the root of a
// region is its smallest pixel index = the flood fill's seed (first pixel in raster order), so
// numbering the roots in index order reproduces the reference's labels (:37,:101).
// =============================================================================================
__device__ __forceinline__ int ccl_find(int32_t* parent, int i) {
    int p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (p != i) {
        i = p;
        p = __hip_atomic_load(&parent[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return i;
}

__device__ __forceinline__ void ccl_union(int32_t* parent, int a, int b) {
    while (true) {
        a = ccl_find(parent, a);
        b = ccl_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }            // hang the larger root under the smaller
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;                                    // b was still a root: joined
        b = old;                                                 // somebody re-parented b meanwhile: retry from there
    }
}

__global__ __launch_bounds__(256) void ccl_init_kernel(int32_t* parent, int32_t* size, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { parent[i] = i; size[i] = 0; }
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(const double* __restrict__ img, int32_t* parent, int W, int H, double maxDiff) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int i = y * W + x;
    const double v = img[i];
    if (isnan(v)) return;
    if (x + 1 < W) {                                             // :53-60 (and :63-70 seen from the other side)
        const double r = img[i + 1];
        if (!isnan(r) && fabs(__dsub_rn(v, r)) < maxDiff) ccl_union(parent, i, i + 1);
    }
    if (y + 1 < H) {                                             // :73-80 / :83-90
        const double b = img[i + W];
        if (!isnan(b) && fabs(__dsub_rn(v, b)) < maxDiff) ccl_union(parent, i, i + W);
    }
}

__global__ __launch_bounds__(256) void ccl_count_kernel(const double* __restrict__ img, int32_t* parent, int32_t* size, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n && !isnan(img[i]);
    int r = -1;
    if (valid) {
        r = ccl_find(parent, i);
        parent[i] = r;                                           // only ever replaces an ancestor by the root
    }
    // :48 regionPixelNum.  Neighbouring pixels mostly share a root, and one big region would otherwise
    // serialise hundreds of thousands of atomics on a single counter: add once per distinct root per wave.
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int lr = __builtin_amdgcn_readlane(r, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && r == lr) & todo;
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&size[lr], __popcll(same));
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void speckle_apply_kernel(const double* __restrict__ img, double* __restrict__ out,
                                                            const int32_t* __restrict__ parent, const int32_t* __restrict__ size,
                                                            int n, double maxSpeckleSize) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double v = img[i];
    const bool drop = !isnan(v) && (double)size[parent[i]] < maxSpeckleSize;      // :94
    out[i] = drop ? __longlong_as_double(0x7FF8000000000000LL) : v;
}

// labels: rank of each region's root among all roots, in index order (3 small kernels: per-block
// counts, a scan of the block counts, ranks)
constexpr int SCAN_CHUNK = 1024;
__device__ __forceinline__ int block_sum_256(int v, int* sh) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const int t = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return t;
}
__global__ __launch_bounds__(256) void roots_count_kernel(const double* __restrict__ img, const int32_t* __restrict__ parent, int32_t* scan, int n) {
    __shared__ int sh[4];
    int c = 0;
    for (int k = 0; k < 4; k++) {
        const int i = blockIdx.x * SCAN_CHUNK + k * 256 + threadIdx.x;
        if (i < n && !isnan(img[i]) && parent[i] == i) c++;
    }
    const int t = block_sum_256(c, sh);
    if (threadIdx.x == 0) scan[blockIdx.x] = t;
}
__global__ void roots_scan_kernel(int32_t* scan, int nb) {       // nb is a few hundred: one thread
    if (threadIdx.x || blockIdx.x) return;
    int acc = 0;
    for (int b = 0; b < nb; b++) { const int t = scan[b]; scan[b] = acc; acc += t; }
}
__global__ __launch_bounds__(256) void roots_rank_kernel(const double* __restrict__ img, const int32_t* __restrict__ parent,
                                                         const int32_t* __restrict__ scan, int32_t* rank, int n) {
    // rank[root] = 1 + number of roots before it; one wave-ordered pass per 256-pixel row of the chunk
    __shared__ int sh[4];
    int base = scan[blockIdx.x];
    for (int k = 0; k < 4; k++) {
        const int i = blockIdx.x * SCAN_CHUNK + k * 256 + threadIdx.x;
        const bool root = i < n && !isnan(img[i]) && parent[i] == i;
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(root);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int before = __popcll(bal & ((1ULL << lane) - 1ULL));
        if (lane == 0) sh[wave] = __popcll(bal);
        __syncthreads();
        int off = 0;
        for (int w = 0; w < wave; w++) off += sh[w];
        const int tot = sh[0] + sh[1] + sh[2] + sh[3];
        if (root) rank[i] = base + off + before + 1;
        base += tot;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void labels_kernel(const double* __restrict__ img, const int32_t* __restrict__ parent,
                                                     const int32_t* __restrict__ rank, int32_t* labels, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    labels[i] = isnan(img[i]) ? 0 : rank[parent[i]];             // :18 zeros, :40/:56 curLabel
}

// =============================================================================================
// vzInd2Disp.m:1-5
// =============================================================================================
__device__ __forceinline__ double vzind2disp(double w, double O, double vMax, double n) {
    const double vzRatio = __dmul_rn(__ddiv_rn(w, n), vMax);
    const double vzInd = __ddiv_rn(vzRatio, __dsub_rn(1.0, vzRatio));
    return __dmul_rn(O, vzInd);
}
__global__ __launch_bounds__(256) void vzind2disp_kernel(const double* __restrict__ w, const double* __restrict__ O,
                                                         double* __restrict__ D, size_t n_px, double vMax, double n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n_px) D[i] = vzind2disp(w[i], O[i], vMax, n);
}

// =============================================================================================
// calc_disp_from_first.m:1-52.  Every pixel offers its value to the four pixels around its target
// (:24-46); a cell starts at -1 (:6) and takes an offer when it holds 0 or something smaller, which
// for maps of non-negative values (vz indices; the precondition of this kernel) is "keep the
// maximum": atomicMax on the bit patterns (non-negative doubles order like integers, -1.0 is a
// negative integer).  A NaN value makes every comparison of :24 false: no offer.
// =============================================================================================
__global__ __launch_bounds__(256) void fill_kernel(double* p, double v, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ __launch_bounds__(256) void disp_from_first_kernel(const double* __restrict__ D1, double* D2, PostGeom g, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t NP = (size_t)W * H, p = (size_t)y * W + x;
    const double v = D1[p];
    const double disp = vzind2disp(v, g.O[p], g.vMax, g.n);                                      // :11
    const double p2x = __dadd_rn(g.Pd0[p], __dmul_rn(disp, g.nd[p]));                            // :13-14
    const double p2y = __dadd_rn(g.Pd0[NP + p], __dmul_rn(disp, g.nd[NP + p]));
    const double sx0 = floor(p2x), sy0 = floor(p2y);                                             // :16
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double sx = sx0 + (double)(k & 1), sy = sy0 + (double)(k >> 1);                    // :17, four corners :24-46
        if (sx >= 1.0 && sx <= (double)W && sy >= 1.0 && sy <= (double)H)
            atomicMax((long long*)&D2[(size_t)((int)sy - 1) * W + ((int)sx - 1)], __double_as_longlong(v));
    }
}

// forward_backward_check.m:1-39: each pixel decides about itself only
__global__ __launch_bounds__(256) void fb_check_map_kernel(const double* __restrict__ D1, const double* __restrict__ D2,
                                                           double* __restrict__ out, PostGeom g, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t NP = (size_t)W * H, p = (size_t)y * W + x;
    const double nan = __longlong_as_double(0x7FF8000000000000LL);
    const double v = D1[p];
    double r = v;
    if (!isnan(v)) {                                                                             // :12
        const double disp = vzind2disp(v, g.O[p], g.vMax, g.n);                                  // :15
        const double p2x = round(__dadd_rn(g.Pd0[p], __dmul_rn(disp, g.nd[p])));                 // :17-20, half away from zero
        const double p2y = round(__dadd_rn(g.Pd0[NP + p], __dmul_rn(disp, g.nd[NP + p])));
        if (!(p2x >= 1.0 && p2x <= (double)W && p2y >= 1.0 && p2y <= (double)H)) r = nan;        // :22 (a NaN target fails every test of :22 and reads D2(NaN): MATLAB errors; here: invalid)
        else {
            const double d2 = D2[(size_t)((int)p2y - 1) * W + ((int)p2x - 1)];
            if (d2 == -1.0 || fabs(__dsub_rn(v, d2)) > 2.0) r = nan;                             // :27,:32 (thr :6)
        }
    }
    out[p] = r;
}

// =============================================================================================
// scanline_in_fill.m:1-70.  Row pass: a run of NaN between two valid pixels takes the smaller of
// the two (:11-22; the run must not touch column 1, :14 -- that case is the left extrapolation),
// runs at the row ends take the nearest valid value (:30-46).  In order-free form: with l / r the
// nearest valid column to the left / right of a NaN pixel in the ORIGINAL row, the result is
// min(v[l], v[r]), v[r] or v[l].  l by a running-maximum scan, r by a running-minimum scan from the
// right (one workgroup per row).  Column pass (:50-69): only the cells above the first / below the
// last valid cell of a column are filled; one thread per column walks it (coalesced across columns).
// =============================================================================================
__device__ __forceinline__ int block_scan_max_256(int v, int* sh) {      // inclusive, in thread order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const int t = __shfl_up(v, s);
        if (lane >= s) v = max(v, t);
    }
    if (lane == 63) sh[wave] = v;
    __syncthreads();
    int pre = INT32_MIN;
    for (int w = 0; w < wave; w++) pre = max(pre, sh[w]);
    __syncthreads();
    return max(v, pre);
}
__global__ __launch_bounds__(256) void infill_rows_kernel(const double* __restrict__ in, double* __restrict__ out, int32_t* __restrict__ left, int W) {
    __shared__ int sh[4];
    __shared__ int carry_sh;
    const size_t row = (size_t)blockIdx.x * W;
    const double* v = in + row;
    int carry = -1;                                              // nearest valid column so far, from the left
    for (int base = 0; base < W; base += 256) {
        const int x = base + threadIdx.x;
        const int mine = (x < W && !isnan(v[x])) ? x : -1;
        const int l = max(block_scan_max_256(mine, sh), carry);
        if (x < W) left[row + x] = l;
        if (threadIdx.x == 255) carry_sh = l;
        __syncthreads();
        carry = carry_sh;
        __syncthreads();
    }
    carry = -1;                                                  // from the right, in mirrored coordinates xr = W-1-x
    for (int base = 0; base < W; base += 256) {
        const int xr = base + threadIdx.x, x = W - 1 - xr;
        const int mine = (xr < W && !isnan(v[x])) ? xr : -1;     // max over xr = min over x
        const int rr = max(block_scan_max_256(mine, sh), carry);
        if (xr < W) {
            const double c = v[x];
            double res = c;
            if (isnan(c)) {
                const int l = left[row + x], r = rr < 0 ? -1 : W - 1 - rr;
                if (l >= 0 && r >= 0) res = fmin(v[l], v[r]);    // :16
                else if (r >= 0) res = v[r];                     // :30-37
                else if (l >= 0) res = v[l];                     // :39-46
            }
            out[row + x] = res;
        }
        if (threadIdx.x == 255) carry_sh = rr;
        __syncthreads();
        carry = carry_sh;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void infill_cols_kernel(double* io, int W, int H) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    int first = -1, last = -1;
    for (int y = 0; y < H; y++)
        if (!isnan(io[(size_t)y * W + x])) { if (first < 0) first = y; last = y; }
    if (first < 0) return;
    const double top = io[(size_t)first * W + x], bot = io[(size_t)last * W + x];
    for (int y = 0; y < first; y++) io[(size_t)y * W + x] = top;            // :52-59
    for (int y = last + 1; y < H; y++) io[(size_t)y * W + x] = bot;         // :61-68
}

// =============================================================================================
// vmf.m:1-14: a 5x5 median per channel (medfilt2, zero padding): the 13th smallest of 25.  One thread
// per pixel; the minimum of the remaining values is removed 12 times, the 13th minimum is the median
// (selection by repeated min/max exchange over a register array: no data-dependent indexing).
// =============================================================================================
__global__ __launch_bounds__(256) void vmf_kernel(const double* __restrict__ in, double* __restrict__ out, int W, int H) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t plane = (size_t)blockIdx.z * W * H;
    double w[25];
#pragma unroll
    for (int dy = -2; dy <= 2; dy++)
#pragma unroll
        for (int dx = -2; dx <= 2; dx++) {
            const int yy = y + dy, xx = x + dx;
            const bool in_img = yy >= 0 && yy < H && xx >= 0 && xx < W;
            w[(dy + 2) * 5 + dx + 2] = in_img ? in[plane + (size_t)(in_img ? yy : 0) * W + (in_img ? xx : 0)] : 0.0;   // zero padding
        }
    // partial selection sort: after pass i, w[i] holds the (i+1)-th smallest
#pragma unroll
    for (int i = 0; i < 13; i++)
#pragma unroll
        for (int j = i + 1; j < 25; j++) {
            const double a = w[i], b = w[j];
            w[i] = fmin(a, b);
            w[j] = fmax(a, b);
        }
    out[plane + (size_t)y * W + x] = w[12];
}

void launch_vmf(hipStream_t st, const double* in, double* out, int W, int H, int channels) {
    hipLaunchKernelGGL(vmf_kernel, dim3((W + 63) / 64, (H + 3) / 4, channels), dim3(256), 0, st, in, out, W, H);
}

// =============================================================================================
// launchers
// =============================================================================================
void launch_speckle_filter(hipStream_t st, const double* image, double* out, int32_t* labels, int32_t* parent,
                           int32_t* size, int32_t* scan, int W, int H, double maxDiff, double maxSpeckleSize) {
    const int n = W * H, nb = (n + 255) / 256;
    hipLaunchKernelGGL(ccl_init_kernel, dim3(nb), dim3(256), 0, st, parent, size, n);
    hipLaunchKernelGGL(ccl_merge_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, image, parent, W, H, maxDiff);
    hipLaunchKernelGGL(ccl_count_kernel, dim3(nb), dim3(256), 0, st, image, parent, size, n);
    hipLaunchKernelGGL(speckle_apply_kernel, dim3(nb), dim3(256), 0, st, image, out, parent, size, n, maxSpeckleSize);
    if (labels) {
        const int nc = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
        hipLaunchKernelGGL(roots_count_kernel, dim3(nc), dim3(256), 0, st, image, parent, scan, n);
        hipLaunchKernelGGL(roots_scan_kernel, dim3(1), dim3(64), 0, st, scan, nc);
        hipLaunchKernelGGL(roots_rank_kernel, dim3(nc), dim3(256), 0, st, image, parent, scan, size, n);    // size reused as rank
        hipLaunchKernelGGL(labels_kernel, dim3(nb), dim3(256), 0, st, image, parent, size, labels, n);
    }
}

void launch_disp_from_first(hipStream_t st, const double* D1, double* D2, const PostGeom& g, int W, int H) {
    const size_t n = (size_t)W * H;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, D2, -1.0, n);   // :6
    hipLaunchKernelGGL(disp_from_first_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, D1, D2, g, W, H);
}

void launch_fb_check(hipStream_t st, const double* D1, const double* D2, double* out, const PostGeom& g, int W, int H) {
    hipLaunchKernelGGL(fb_check_map_kernel, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, D1, D2, out, g, W, H);
}

void launch_scanline_in_fill(hipStream_t st, const double* in, double* out, int32_t* left, int W, int H) {
    hipLaunchKernelGGL(infill_rows_kernel, dim3(H), dim3(256), 0, st, in, out, left, W);
    hipLaunchKernelGGL(infill_cols_kernel, dim3((W + 255) / 256), dim3(256), 0, st, out, W, H);
}

void launch_vzind2disp(hipStream_t st, const double* w, const double* O, double* D, size_t n_px, double vMax, double n) {
    hipLaunchKernelGGL(vzind2disp_kernel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, st, w, O, D, n_px, vMax, n);
}

}  // namespace fsgm
