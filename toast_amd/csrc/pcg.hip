// pcg.hip -- the scalars of the preconditioned conjugate gradient kept on the device.
//
// The reference's solver (src/toast/ops/mapmaker_solve.py:524-755) computes three dot products per iteration on the
// host -- p.Ap for the step length, r.r for the convergence test, z.r for the direction update -- and each of them is a
// device -> host round trip when the amplitude vectors live on the GPU: the host cannot enqueue the next kernel before
// the previous scalar has arrived, ~30 % of an iteration at configs[1] size.  Here the recurrence's scalars and its
// control flow (convergence test, the stall test every 10 iterations, the iteration limit) live in one small device
// structure; the dot products reduce into it, one-thread "stage" kernels do the scalar arithmetic in the reference's
// order, and the vector updates read alpha / beta from it.  The host only enqueues; it learns about convergence from
// an asynchronous copy of the status that it reads ONE ITERATION LATE, so the GPU never waits for it.  An iteration
// enqueued after the solver finished is a no-op for the solution: alpha = 0 leaves result and residual bit for bit as
// they were (x + 0 * p == x), the direction update is skipped.
//
// Several processes: the local dot product is summed over the ranks by the library's communicator on the same stream
// (toast_hip_comm, comm.cpp) before the stage kernel reads it.
#include "kernel_common.hpp"

#include "../../include/toast_hip.h"

namespace {

struct PcgState {
    double delta, p_ap, alpha, neg_alpha, sqsum, sqsum_init, sqsum_best, last_best, beta, live, tmp, convergence,
        relative;
    int64_t it, n_iter_min, n_iter_max, done, n_history;
    unsigned int ticket, pad;    // blocks of the running dot product that have stored their partial sum
    toast_hip_pcg_status stat;   // refreshed by every stage for the asynchronous copy to the host
    double history[1];           // n_iter_max entries follow
};

constexpr int kDotBlocks = 1024;    // partial sums per dot product (the scratch block holds that many)
constexpr int kDotPer = 8;          // elements per thread and pass

// what a dot-product launch does to the vectors on its way (the sum is always over the values it leaves behind):
enum { kDotPlain = 0, kDotStep = 1, kDotPrecond = 2 };

struct DotFused {
    const double * p;        // kDotStep: proposal
    double * result;         //           result += alpha p
    const double * ap;       //           x (= residual) += -alpha ap, then x . x
    const double * var;      // kDotPrecond: x (= z) = y (= residual) * var where y is unflagged, else 0, then x . y
};

__global__ void k_pcg_init(PcgState * __restrict__ s, double sqsum_init, double delta, double convergence,
                           int64_t n_iter_min, int64_t n_iter_max) {
    s->delta = delta;
    s->p_ap = 0.0;
    s->alpha = 0.0;
    s->neg_alpha = 0.0;
    s->sqsum = sqsum_init;
    s->sqsum_init = sqsum_init;
    s->sqsum_best = sqsum_init;
    s->last_best = sqsum_init;
    s->beta = 1.0;
    s->live = (n_iter_max > 0) ? 1.0 : 0.0;
    s->tmp = 0.0;
    s->convergence = convergence;
    s->relative = 0.0;
    s->it = 0;
    s->n_iter_min = n_iter_min;
    s->n_iter_max = n_iter_max;
    s->done = (n_iter_max > 0) ? 0 : TOAST_HIP_PCG_MAX_ITER;
    s->n_history = 0;
    s->ticket = 0;
    s->pad = 0;
    s->stat.iteration = 0;
    s->stat.done = s->done;
    s->stat.n_history = 0;
    s->stat.relative = 0.0;
    s->stat.sqsum = sqsum_init;
}

// The scalar part of one third of an iteration (reference mapmaker_solve.py:672-745), `tmp` = the dot product that was
// just reduced:  stage 1  alpha = delta / (p . A p);   stage 2  sqsum = r . r, history, convergence and stall tests;
// stage 3  beta = (z . r) / delta, delta = z . r, next iteration.  One thread.
__device__ void pcg_stage(PcgState * __restrict__ s, int stage) {
    if (s->done != 0) {
        s->alpha = 0.0;
        s->neg_alpha = 0.0;
        s->beta = 1.0;
        s->live = 0.0;
    } else if (stage == 1) {
        s->p_ap = s->tmp;
        s->alpha = s->delta / s->p_ap;
        s->neg_alpha = -s->alpha;
    } else if (stage == 2) {
        const double sqsum = s->tmp;
        s->sqsum = sqsum;
        const double relative = (s->sqsum_init != 0.0) ? sqsum / s->sqsum_init : 0.0;
        s->relative = relative;
        s->history[s->it] = relative;
        s->n_history = s->it + 1;
        if (!isfinite(sqsum)) {
            s->done = TOAST_HIP_PCG_NOT_FINITE;
        } else if (relative < s->convergence || sqsum < 1.0e-30) {
            s->done = TOAST_HIP_PCG_CONVERGED;
        } else {
            s->sqsum_best = (sqsum < s->sqsum_best) ? sqsum : s->sqsum_best;
            if (s->it % 10 == 0 && s->it >= s->n_iter_min) {
                if (s->last_best < s->sqsum_best * 2.0) {
                    s->done = TOAST_HIP_PCG_STALLED;
                } else {
                    s->last_best = s->sqsum_best;
                }
            }
        }
        if (s->done != 0) s->live = 0.0;
    } else {
        const double delta_last = s->delta;
        s->delta = s->tmp;
        s->beta = s->delta / delta_last;
        s->it += 1;
        if (s->it >= s->n_iter_max) s->done = TOAST_HIP_PCG_MAX_ITER;   // (this iteration's direction update still runs)
    }
    s->stat.iteration = s->it;
    s->stat.done = s->done;
    s->stat.n_history = s->n_history;
    s->stat.relative = s->relative;
    s->stat.sqsum = s->sqsum;
}

__global__ void k_pcg_stage(PcgState * __restrict__ s, int stage) { pcg_stage(s, stage); }

// Dot product, reduction and stage in ONE launch (a launch costs ~5 us of device time, an iteration at configs[1] size
// 400): every block stores its partial sum and takes a ticket; the block that draws the last ticket adds the partials
// in block order (the result does not depend on which block that is) and runs the stage.
// MODE kDotStep / kDotPrecond: the vector update whose OUTPUT the dot product reads runs in the same pass (result and
// residual update before r . r; diagonal preconditioner before z . r) -- two launches and one sweep over the vectors
// less per iteration each.  Element -> thread assignment and summation order are the same in all modes, so the fused
// forms give the bits of update kernel + plain dot product.
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_pcg_dot_stage(int64_t n, double * __restrict__ x,
                                                            const double * __restrict__ y,
                                                            const uint8_t * __restrict__ fx,
                                                            const uint8_t * __restrict__ fy, DotFused f,
                                                            double * __restrict__ partials, PcgState * __restrict__ s,
                                                            int accumulate, int stage) {
    __shared__ double s_part[kThreads / 64];
    __shared__ bool s_last;
    // eight elements per thread and pass, all loads of a pass issued together (a plain grid-stride loop waits for one
    // element after the other: 34 us for 230 k amplitudes at 32 per thread)
    constexpr int kPer = kDotPer;
    const double a = (MODE == kDotStep) ? s->alpha : 0.0, na = (MODE == kDotStep) ? s->neg_alpha : 0.0;
    double acc = 0.0;
    for (int64_t base = (int64_t)blockIdx.x * kThreads * kPer; base < n; base += (int64_t)gridDim.x * kThreads * kPer) {
        double xv[kPer], yv[kPer];
        bool good[kPer];
        if constexpr (MODE == kDotStep) {
            double pv[kPer], rv[kPer], av[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int64_t i = base + threadIdx.x + (int64_t)k * kThreads;
                const int64_t j = (i < n) ? i : 0;
                xv[k] = x[j];
                av[k] = f.ap[j];
                pv[k] = f.p[j];
                rv[k] = f.result[j];
                good[k] = (i < n) && (fx == nullptr || fx[j] == 0);
            }
            if (a != 0.0) {      // (alpha = 0: the solver has finished, both vectors stay bit for bit)
#pragma unroll
                for (int k = 0; k < kPer; ++k) {
                    const int64_t i = base + threadIdx.x + (int64_t)k * kThreads;
                    xv[k] = xv[k] + na * av[k];
                    if (i < n) {
                        f.result[i] = rv[k] + a * pv[k];
                        x[i] = xv[k];
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kPer; ++k) yv[k] = xv[k];
        } else if constexpr (MODE == kDotPrecond) {
            double vv[kPer];
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int64_t i = base + threadIdx.x + (int64_t)k * kThreads;
                const int64_t j = (i < n) ? i : 0;
                yv[k] = y[j];
                vv[k] = f.var[j];
                const bool unflagged = (fy[j] == 0);
                good[k] = (i < n) && unflagged && (fx == nullptr || fx[j] == 0);
                xv[k] = unflagged ? yv[k] * vv[k] : 0.0;
                if (i < n) x[i] = xv[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kPer; ++k) {
                const int64_t i = base + threadIdx.x + (int64_t)k * kThreads;
                const bool in = i < n;
                const int64_t j = in ? i : 0;
                xv[k] = x[j];
                yv[k] = y[j];
                good[k] = in && (fx == nullptr || fx[j] == 0) && (fy == nullptr || fy[j] == 0);
            }
        }
#pragma unroll
        for (int k = 0; k < kPer; ++k) {
            if (good[k]) acc += xv[k] * yv[k];
        }
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) t += s_part[w];
        __hip_atomic_store(&partials[blockIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned int ticket = atomicAdd(&s->ticket, 1u);
        s_last = (ticket == gridDim.x - 1);
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (threadIdx.x < 64) {
        double t = 0.0;
        for (int b = threadIdx.x; b < (int)gridDim.x; b += 64) {
            t += __hip_atomic_load(&partials[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
        if (threadIdx.x == 0) {
            s->tmp = (accumulate ? s->tmp : 0.0) + t;
            s->ticket = 0;
            if (stage != 0) pcg_stage(s, stage);
        }
    }
}

__device__ __forceinline__ double pcg_scalar(const PcgState * s, int sel) {
    switch (sel) {
        case TOAST_HIP_PCG_ALPHA: return s->alpha;
        case TOAST_HIP_PCG_NEG_ALPHA: return s->neg_alpha;
        case TOAST_HIP_PCG_BETA: return s->beta;
        case TOAST_HIP_PCG_LIVE: return s->live;
        default: return 1.0;
    }
}

__global__ __launch_bounds__(kThreads) void k_pcg_axpby(const PcgState * __restrict__ s, int64_t n, int a_sel,
                                                        const double * __restrict__ x, int b_sel,
                                                        double * __restrict__ y) {
    const double a = pcg_scalar(s, a_sel), b = pcg_scalar(s, b_sel);
    if (a == 0.0 && b == 1.0) return;    // the solver has finished: y stays bit for bit
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        y[i] = (b == 1.0) ? y[i] + a * x[i] : a * x[i] + b * y[i];
    }
}

// result += alpha proposal and residual -= alpha lhs_out in one launch
__global__ __launch_bounds__(kThreads) void k_pcg_step(const PcgState * __restrict__ s, int64_t n,
                                                       const double * __restrict__ p, double * __restrict__ result,
                                                       const double * __restrict__ ap, double * __restrict__ residual) {
    const double a = s->alpha, na = s->neg_alpha;
    if (a == 0.0) return;    // the solver has finished: both vectors stay bit for bit
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
        result[i] = result[i] + a * p[i];
        residual[i] = residual[i] + na * ap[i];
    }
}

struct StatusRing {
    static constexpr int kSlots = 4;
    toast_hip_pcg_status * host = nullptr;   // page-locked
    hipEvent_t ev[kSlots];
    bool pending[kSlots] = {false, false, false, false};
    int64_t next = 0;
    const void * owner = nullptr;            // the state block whose solver runs ahead through this ring
};

StatusRing & ring() {
    static StatusRing r;
    if (r.host == nullptr) {
        TH_HIP(hipHostMalloc(reinterpret_cast<void **>(&r.host), sizeof(toast_hip_pcg_status) * StatusRing::kSlots,
                             hipHostMallocDefault));
        for (int i = 0; i < StatusRing::kSlots; ++i) TH_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
    }
    return r;
}

}  // namespace

extern "C" {

int toast_hip_pcg_state_bytes(int64_t n_iter_max, size_t * bytes) {
    return guarded([&] {
        if (n_iter_max < 0) fail_arg("n_iter_max must not be negative");
        *bytes = sizeof(PcgState) + sizeof(double) * (size_t)n_iter_max;
    });
}

int toast_hip_pcg_init_dev(void * d_state, double sqsum_init, double delta, double convergence, int64_t n_iter_min,
                           int64_t n_iter_max, void * stream) {
    return guarded([&] {
        hipLaunchKernelGGL(k_pcg_init, dim3(1), dim3(1), 0, as_stream(stream), static_cast<PcgState *>(d_state),
                           sqsum_init, delta, convergence, n_iter_min, n_iter_max);
        check_launch();
        StatusRing & r = ring();
        for (int i = 0; i < StatusRing::kSlots; ++i) r.pending[i] = false;
        r.next = 0;
        r.owner = d_state;      // (the solver initialised last; any other state block reads its status synchronously)
    });
}

static void dot_launch(int mode, void * d_state, int64_t n, double * d_x, const double * d_y, const uint8_t * d_fx,
                       const uint8_t * d_fy, const DotFused & f, int accumulate, int stage, void * stream) {
    if (stage < 0 || stage > 3) fail_arg("pcg stage must be 0 (none), 1, 2 or 3");
    double * d_part = (double *)Manager::get().scratch(Manager::kScratchDot, sizeof(double) * 1032) + 8;
    hipStream_t st = as_stream(stream);
    // few blocks: every block ends in one atomic on the same ticket, and the last one adds all partial sums
    // (one element per thread, 900 blocks: 26 us per dot product at configs[1] size).  Eight elements per thread and
    // pass; long vectors get up to kDotBlocks blocks (3.7 M amplitudes with 256 blocks = 7 passes of one 4-wave block
    // per CU: 67 us for 59 MB)
    int64_t nb = (n + (int64_t)kThreads * kDotPer - 1) / ((int64_t)kThreads * kDotPer);
    if (nb < 1) nb = 1;
    if (nb > kDotBlocks) nb = kDotBlocks;
    const dim3 grid((unsigned)nb);
    PcgState * s = static_cast<PcgState *>(d_state);
    if (mode == kDotStep) {
        hipLaunchKernelGGL(k_pcg_dot_stage<kDotStep>, grid, dim3(kThreads), 0, st, n, d_x, d_y, d_fx, d_fy, f, d_part, s,
                           accumulate, stage);
    } else if (mode == kDotPrecond) {
        hipLaunchKernelGGL(k_pcg_dot_stage<kDotPrecond>, grid, dim3(kThreads), 0, st, n, d_x, d_y, d_fx, d_fy, f, d_part,
                           s, accumulate, stage);
    } else {
        hipLaunchKernelGGL(k_pcg_dot_stage<kDotPlain>, grid, dim3(kThreads), 0, st, n, d_x, d_y, d_fx, d_fy, f, d_part, s,
                           accumulate, stage);
    }
    check_launch();
}

int toast_hip_pcg_dot_dev(void * d_state, int64_t n, const double * d_x, const double * d_y, const uint8_t * d_flags_x,
                          const uint8_t * d_flags_y, int accumulate, int stage, void * stream) {
    return guarded([&] {
        dot_launch(kDotPlain, d_state, n, const_cast<double *>(d_x), d_y, d_flags_x, d_flags_y, DotFused{}, accumulate,
                   stage, stream);
    });
}

int toast_hip_pcg_step_dot_dev(void * d_state, int64_t n, const double * d_proposal, double * d_result,
                               const double * d_lhs_out, double * d_residual, const uint8_t * d_flags, int accumulate,
                               int stage, void * stream) {
    return guarded([&] {
        DotFused f{};
        f.p = d_proposal;
        f.result = d_result;
        f.ap = d_lhs_out;
        dot_launch(kDotStep, d_state, n, d_residual, d_residual, d_flags, d_flags, f, accumulate, stage, stream);
    });
}

int toast_hip_pcg_precond_diag_dot_dev(void * d_state, int64_t n, const double * d_var, const double * d_residual,
                                       const uint8_t * d_flags_residual, double * d_out, const uint8_t * d_flags_out,
                                       int accumulate, int stage, void * stream) {
    return guarded([&] {
        if (d_flags_residual == nullptr) fail_arg("the diagonal preconditioner needs the amplitude flags");
        DotFused f{};
        f.var = d_var;
        dot_launch(kDotPrecond, d_state, n, d_out, d_residual, d_flags_out, d_flags_residual, f, accumulate, stage,
                   stream);
    });
}

int toast_hip_pcg_step_dev(const void * d_state, int64_t n, const double * d_proposal, double * d_result,
                           const double * d_lhs_out, double * d_residual, void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_pcg_step, flat_grid(n), dim3(kThreads), 0, as_stream(stream),
                           static_cast<const PcgState *>(d_state), n, d_proposal, d_result, d_lhs_out, d_residual);
        check_launch();
    });
}

int toast_hip_pcg_stage_dev(void * d_state, int stage, int allreduce, void * stream) {
    return guarded([&] {
        if (stage < 1 || stage > 3) fail_arg("pcg stage must be 1, 2 or 3");
        PcgState * s = static_cast<PcgState *>(d_state);
        if (allreduce) {
            const int rc = toast_hip_comm_allreduce_dev(&s->tmp, 1, TOAST_HIP_COMM_F64, TOAST_HIP_COMM_SUM, stream);
            if (rc != 0) throw Error(rc, toast_hip_last_error());
        }
        hipLaunchKernelGGL(k_pcg_stage, dim3(1), dim3(1), 0, as_stream(stream), s, stage);
        check_launch();
    });
}

int toast_hip_pcg_axpby_dev(const void * d_state, int64_t n, int a_sel, const double * d_x, int b_sel, double * d_y,
                            void * stream) {
    return guarded([&] {
        if (n <= 0) return;
        hipLaunchKernelGGL(k_pcg_axpby, flat_grid(n), dim3(kThreads), 0, as_stream(stream),
                           static_cast<const PcgState *>(d_state), n, a_sel, d_x, b_sel, d_y);
        check_launch();
    });
}

int toast_hip_pcg_status_dev(void * d_state, int lag, toast_hip_pcg_status * out, void * stream) {
    return guarded([&] {
        if (lag < 0 || lag >= StatusRing::kSlots) fail_arg("pcg status lag must be 0 .. 3");
        StatusRing & r = ring();
        hipStream_t st = as_stream(stream);
        PcgState * s = static_cast<PcgState *>(d_state);
        if (r.owner != d_state) {
            // one ring per process: a second solver interleaved with the one that owns it gets the current status
            copy_to_host(out, &s->stat, sizeof(toast_hip_pcg_status), st);
            return;
        }
        const int slot = (int)(r.next % StatusRing::kSlots);
        TH_HIP(hipMemcpyAsync(&r.host[slot], &s->stat, sizeof(toast_hip_pcg_status), hipMemcpyDeviceToHost, st));
        TH_HIP(hipEventRecord(r.ev[slot], st));
        r.pending[slot] = true;
        const int64_t want = r.next - lag;
        r.next += 1;
        if (want < 0) {
            out->iteration = 0;
            out->done = 0;
            out->n_history = 0;
            out->relative = 0.0;
            out->sqsum = 0.0;
            return;
        }
        const int ws = (int)(want % StatusRing::kSlots);
        TH_HIP(hipEventSynchronize(r.ev[ws]));
        *out = r.host[ws];
    });
}

int toast_hip_pcg_history_dev(void * d_state, double * history, int64_t capacity, toast_hip_pcg_status * final_status,
                              void * stream) {
    return guarded([&] {
        hipStream_t st = as_stream(stream);
        PcgState * s = static_cast<PcgState *>(d_state);
        PcgState head;
        copy_to_host(&head, s, sizeof(PcgState), st);      // (through the page-locked bounce ring; synchronises)
        if (final_status != nullptr) {
            final_status->iteration = head.it;
            final_status->done = head.done;
            final_status->n_history = head.n_history;
            final_status->relative = head.relative;
            final_status->sqsum = head.sqsum;
        }
        const int64_t n = (head.n_history < capacity) ? head.n_history : capacity;
        if (n > 0 && history != nullptr) {
            copy_to_host(history, reinterpret_cast<const char *>(s) + offsetof(PcgState, history), sizeof(double) * n, st);
        }
    });
}

}  // extern "C"
