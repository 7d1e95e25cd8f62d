// What the translation units behind the C ABI share: the plan handle, its per-device
// state and the launch path.  Not installed; include/ctc_tc.h is the public face.
#pragma once

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "ctc_tc.h"
#include "kernels.hpp"
#include "planner.hpp"

namespace ctc_internal {

// offset tables of one plan resident on one device
struct DeviceTables {
    int64_t *base = nullptr;
    const int64_t *row_a = nullptr, *k_a = nullptr, *k_b = nullptr, *col_b = nullptr;
    const int64_t *row_c = nullptr, *col_c = nullptr, *bat_a = nullptr, *bat_b = nullptr, *bat_c = nullptr;
    const int64_t *sk[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // bx cx by cy bo co
    // a batch of small row-major products (see small_batched in api.cpp): one descriptor per batch entry
    ctc::SmallGemm *small = nullptr;
    bool small_a_outer = false, small_b_kmajor = false;  // how the descriptors' operands are stored (ctc::launch_grouped)
    // workspace of this plan on this device (split-K partial sums, or the three packed tensors of the
    // explicit-transpose path): grows on demand; executions on different streams are ordered through
    // ws_done so that they never share it concurrently
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    // workspaces this one outgrew: kernels in flight may still use them, so they are kept until the
    // plan is destroyed instead of synchronising the device inside an execution call
    std::vector<void *> retired;
};

int set_last_error(int code, const std::string &msg);
int set_last_hip_error(hipError_t e, const char *what);

// No C++ exception leaves the C ABI: the body of an entry point that allocates host memory (plans, offset
// tables, strings) stands between these two.
#define CTC_ABI_TRY try {
#define CTC_ABI_CATCH                                                                                           \
    }                                                                                                           \
    catch (const std::bad_alloc &) { return ctc_internal::set_last_error(CTC_TC_ENOMEM, "out of host memory"); } \
    catch (const std::exception &e)                                                                             \
    {                                                                                                           \
        return ctc_internal::set_last_error(CTC_TC_EINVAL, std::string("internal error: ") + e.what());         \
    }

}  // namespace ctc_internal

struct ctc_tc_plan {
    ctc::Plan plan;
    std::mutex mu;
    std::map<int, ctc_internal::DeviceTables> tables;  // by device ordinal
    std::unique_ptr<ctc_tc_plan> packed;  // GEMM on packed [batch,M,K] x [batch,K,N] operands (TTGT)
    // GEMM on K-major rows [batch,M,Kp] x [batch,N,Kp] (Kp = Plan::k_padded, one fused contracted mode), C in
    // place; perm_bn: the transpose of B into [batch,N,K]
    std::unique_ptr<ctc_tc_plan> padded;
    std::vector<int> perm_bn;
    std::map<int, const void *> pads_zeroed;  // by device: the workspace whose pad columns have been cleared
    std::mutex transposes_mu;             // one execution through the internal transpose workspace at a time
};

namespace ctc_internal {

// Where the split-K partial sums of a launch go.
struct Workspace {
    void *fixed = nullptr;   // non-null: use this buffer (of fixed_bytes) and never allocate
    size_t fixed_bytes = 0;
};

// Enqueues the plan's kernel(s) on `stream`.  With ws == nullptr the plan's own workspace is used
// (allocated or grown on demand); with a fixed workspace nothing is allocated and a launch that
// needs more than fixed_bytes fails with CTC_TC_EUNSUPPORTED.  no_alloc additionally requires the
// plan's tables to be resident on the device already (ctc_tc_plan_prepare).
int launch_plan(ctc_tc_plan *pl, double alpha, const void *A, const void *B, double beta, void *C,
                hipStream_t stream, const Workspace *ws = nullptr, bool no_alloc = false);

}  // namespace ctc_internal
