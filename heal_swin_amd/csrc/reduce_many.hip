// Deferred, batched partial-sum reductions (hs_reduce_flush).
//
// Every split-K weight-gradient launch (csrc/linear_wgrad.hip) and every LayerNorm backward (csrc/layernorm.hip) ends in a
// tiny "sum the per-workgroup partial records into the parameter's fp32 gradient" launch: 239 + 100 of them per HEAL-SWIN-B step,
// 147 + 52 per HEAL-SWIN-T step -- each 5-11 us of GPU time that is launch ramp and a serial walk over the slices, 6 % + 1.7 % of
// the T@128 step (profiles/archive_r01_r04/r04_k_T128_summary.txt).  Nothing on the backward's critical path reads a parameter gradient, so
// callers that deposit into gradient buffers may pass HS_ACC_DEFER in `accumulate`: the producing kernel runs as before, the
// reduction is QUEUED (host side, per thread) and hs_reduce_flush folds all queued jobs in ONE launch -- before the gradient
// exchange of a bucket, at the end of the backward pass, or when the queue is full.  The partial records must stay alive (and
// unwritten) until the flush; the Python binding keeps the workspaces referenced (ops._defer_keep).
//
// One job = `slices` records of `count` floats at a stride, summed record-wise in a FIXED order into two destinations (the first
// n_w floats -> dw, the rest -> db): exactly the layout both producers already write.  A workgroup of 256 threads owns
// 256 / PH float4 columns of one job and PH row phases: thread (column, phase) adds records phase, phase + PH, ... in order, the
// phases are combined through LDS in order -- deterministic, no atomics; PH follows the record count so that short stacks keep
// one thread per column and tall ones (170 slices of a 96 x 288 gradient; 2048 LayerNorm partial rows) are walked 16 rows at a time.
#include <mutex>
#include <unordered_map>

#include "hs_device.h"

namespace hs {

struct ReduceJob {
    const float* part;
    float* dw;
    float* db;
    int64_t in_stride;
    int64_t n_w;
    int64_t count;
    int slices;
    int accumulate;
    int phases;      // 1, 4, 16 or 64
    int first_block;
};

constexpr int kMaxJobs = 44;  // 44 x 64 B + header < 4 KB of kernel arguments

struct ReduceTable {
    int n;
    int total_blocks;
    ReduceJob job[kMaxJobs];
};

namespace {

// host-side queues, one per stream: the producers run on autograd's worker thread, the flush at the end of the pass on the
// caller's thread -- the stream, not the thread, is what orders a job's producer, the flush and the consumers of the gradient
std::mutex g_mutex;
std::unordered_map<hipStream_t, ReduceTable> g_queues;

__global__ void __launch_bounds__(256) reduce_many_kernel(const ReduceTable t) {
    __shared__ float4 red[256];
    int j = 0;
    while (j + 1 < t.n && (int)blockIdx.x >= t.job[j + 1].first_block) ++j;  // uniform: a handful of scalar compares
    const ReduceJob& q = t.job[j];
    const int ph = q.phases, cols = 256 / ph;
    const int col = threadIdx.x % cols, phase = threadIdx.x / cols;
    const int64_t e = ((int64_t)(blockIdx.x - q.first_block) * cols + col) * 4;  // n_w and count are multiples of 4
    const bool live = e < q.count;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) {
#pragma unroll 8
        for (int s = phase; s < q.slices; s += ph) {
            const float4 v = *(const float4*)(q.part + (int64_t)s * q.in_stride + e);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (ph > 1) {
        red[threadIdx.x] = acc;
        __syncthreads();
        if (phase == 0) {
            for (int p = 1; p < ph; ++p) {
                const float4 v = red[p * cols + col];
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    }
    if (!live || phase != 0) return;
    float* dst;
    if (e < q.n_w) dst = q.dw + e;
    else if (q.db) dst = q.db + (e - q.n_w);
    else return;
    if (q.accumulate) {
        const float4 o = *(const float4*)dst;
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    *(float4*)dst = acc;
}

// row phases per record stack: one thread per column for short stacks, up to 64 rows walked side by side for tall ones (the 2048
// partial rows of a LayerNorm backward: 64 workgroups x 32 rows per thread, as the dedicated kernel this replaced)
int reduce_phases(int slices) { return slices <= 8 ? 1 : (slices <= 64 ? 4 : (slices <= 256 ? 16 : 64)); }

// Does [dw, dw + n_w) or [db, db + count - n_w) overlap a destination of a pending job?
bool overlaps_pending(const ReduceTable& t, const float* dw, int64_t n_w, const float* db, int64_t count) {
    const float* dw_end = dw + n_w;
    const float* db_end = db ? db + (count - n_w) : nullptr;
    auto hit = [](const float* a0, const float* a1, const float* b0, const float* b1) { return a0 && b0 && a0 < b1 && b0 < a1; };
    for (int i = 0; i < t.n; ++i) {
        const ReduceJob& p = t.job[i];
        const float* pw_end = p.dw + p.n_w;
        const float* pb_end = p.db ? p.db + (p.count - p.n_w) : nullptr;
        if (hit(dw, dw_end, p.dw, pw_end) || hit(dw, dw_end, p.db, pb_end) || hit(db, db_end, p.dw, pw_end) ||
            hit(db, db_end, p.db, pb_end))
            return true;
    }
    return false;
}

}  // namespace

static int flush_locked(hipStream_t s) {
    auto it = g_queues.find(s);
    if (it == g_queues.end() || it->second.n == 0) return HS_OK;
    ReduceTable& t = it->second;
    hipLaunchKernelGGL(reduce_many_kernel, dim3((unsigned)t.total_blocks), dim3(256), 0, s, t);
    t.n = 0;
    t.total_blocks = 0;
    HS_LAUNCH_CHECK("reduce_many");
    return HS_OK;
}

int reduce_flush(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mutex);
    return flush_locked(s);
}

int reduce_pending(hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_queues.find(s);
    return it == g_queues.end() ? 0 : it->second.n;
}

// Queue (or, when the queue is full or holds a job with an overlapping destination, flush first and then queue) one reduction; `s` is the stream of the producing launch.
int reduce_defer(const float* part, int64_t in_stride, int slices, int64_t n_w, int64_t count, float* dw, float* db, int accumulate,
                 hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mutex);
    ReduceTable& t = g_queues[s];
    // The jobs of one flush run side by side in ONE launch and read-modify-write their destinations without atomics: two pending
    // jobs must never share a destination element (the three bf16x3 products of an fp32 weight gradient, a parameter used twice
    // in one backward, micro-batches accumulated without a flush in between).  A job that overlaps a pending one therefore
    // flushes the queue first -- the earlier sum is then stream-ordered before it, as with immediate reductions.
    const bool clash = t.n == kMaxJobs || overlaps_pending(t, dw, n_w, db, count);
    if (clash) {
        if (int st = flush_locked(s)) return st;
    }
    ReduceJob& q = t.job[t.n];
    q.part = part;
    q.dw = dw;
    q.db = db;
    q.in_stride = in_stride;
    q.n_w = n_w;
    q.count = count;
    q.slices = slices;
    q.accumulate = accumulate ? 1 : 0;
    q.phases = reduce_phases(slices);
    q.first_block = t.total_blocks;
    const int cols = 256 / q.phases;
    t.total_blocks += (int)((count / 4 + cols - 1) / cols);
    ++t.n;
    return HS_OK;
}

// The same sum launched at once (a one-job table): the immediate and the deferred form of a reduction share kernel and summation
// order, so a gradient does not depend on whether its sum was queued.
int reduce_now(const float* part, int64_t in_stride, int slices, int64_t n_w, int64_t count, float* dw, float* db, int accumulate,
               hipStream_t s) {
    {  // a queued sum into the same destination must land first (stream order = issue order, as without deferral)
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_queues.find(s);
        if (it != g_queues.end() && it->second.n && overlaps_pending(it->second, dw, n_w, db, count)) {
            if (int st = flush_locked(s)) return st;
        }
    }
    ReduceTable t{};
    ReduceJob& q = t.job[0];
    q.part = part;
    q.dw = dw;
    q.db = db;
    q.in_stride = in_stride;
    q.n_w = n_w;
    q.count = count;
    q.slices = slices;
    q.accumulate = accumulate ? 1 : 0;
    q.phases = reduce_phases(slices);
    q.first_block = 0;
    const int cols = 256 / q.phases;
    t.n = 1;
    t.total_blocks = (int)((count / 4 + cols - 1) / cols);
    hipLaunchKernelGGL(reduce_many_kernel, dim3((unsigned)t.total_blocks), dim3(256), 0, s, t);
    HS_LAUNCH_CHECK("reduce_now");
    return HS_OK;
}

}  // namespace hs

extern "C" {

int hs_reduce_pending(void* stream) { return hs::reduce_pending((hipStream_t)stream); }

int hs_reduce_flush(void* stream) { return hs::reduce_flush((hipStream_t)stream); }

}  // extern "C"
