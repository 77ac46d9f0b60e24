// Pre-solve row hygiene on the device: the reference's removeNullConstraints(J, c, threshold)
// (HelperFuncs.py:41-54, called at CollocationBasedPlanner2D.py:338 before the equality rows reach solvers.qp):
// rows of [J | c] whose coefficients are ALL below `threshold` in magnitude are dropped, the others keep their order.
// Three small launches: per-row flag (one wave per row), exclusive scan of the flags (one workgroup), ordered copy.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void null_row_flag_kernel(const double* J, int rows, int cols, double thr, int* keep) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const double* r = J + (size_t)row * cols;
    bool any = false;
    for (int j = lane; j < cols; j += 64) any = any || !(fabs(r[j]) < thr);     // NaN counts as a coefficient, as in the reference
    const bool k = __any(any);
    if (lane == 0) keep[row] = k ? 1 : 0;
}

// pos[i] = number of kept rows before row i; counts[0] = rows kept, counts[1] = dropped rows whose rhs is not null
__global__ __launch_bounds__(1024) void null_row_scan_kernel(const int* keep, const double* c, int rows, double thr, int* pos,
                                                             int* counts) {
    __shared__ int part[1024];
    __shared__ int carry, odd;
    const int tid = threadIdx.x;
    if (tid == 0) { carry = 0; odd = 0; }
    __syncthreads();
    for (int base = 0; base < rows; base += 1024) {
        const int i = base + tid;
        const int k = i < rows ? keep[i] : 0;
        if (i < rows && !k && fabs(c[i]) > thr) atomicAdd(&odd, 1);
        part[tid] = k;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {            // inclusive Hillis-Steele scan of the chunk
            const int v = tid >= s ? part[tid - s] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (i < rows) pos[i] = carry + part[tid] - k;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) { counts[0] = carry; counts[1] = odd; }
}

__global__ __launch_bounds__(256) void null_row_copy_kernel(const double* J, const double* c, int rows, int cols, const int* keep,
                                                            const int* pos, double* Jo, double* co) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows || !keep[row]) return;
    const double* r = J + (size_t)row * cols;
    double* o = Jo + (size_t)pos[row] * cols;
    for (int j = lane; j < cols; j += 64) o[j] = r[j];
    if (lane == 0) co[pos[row]] = c[row];
}

}  // namespace

extern "C" int mpcqp_remove_null_rows(int device_id, int rows, int cols, const double* J, const double* c, double threshold,
                                      double* J_out, double* c_out, int* counts, int* int_ws, void* stream) {
    if (rows < 0 || cols < 1 || !counts || (rows > 0 && (!J || !c || !J_out || !c_out || !int_ws))) return MPCQP_EINVAL;
    if (rows > 0 && (J_out == J || c_out == c)) return MPCQP_EINVAL;        // the ordered copy is not in place
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rows == 0) return hipMemsetAsync(counts, 0, 2 * sizeof(int), st) == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
    int* keep = int_ws; int* pos = int_ws + rows;
    hipLaunchKernelGGL(null_row_flag_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, J, rows, cols, threshold, keep);
    hipLaunchKernelGGL(null_row_scan_kernel, dim3(1), dim3(1024), 0, st, keep, c, rows, threshold, pos, counts);
    hipLaunchKernelGGL(null_row_copy_kernel, dim3(ceil_div(rows, 4)), dim3(256), 0, st, J, c, rows, cols, keep, pos, J_out, c_out);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
