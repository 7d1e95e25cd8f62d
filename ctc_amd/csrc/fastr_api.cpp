// C ABI of the FASTR scoring path (include/ctc_fastr.h): builds the work table from the
// per-region counts on the host, uploads it through a pinned staging buffer and launches.
#include "ctc_fastr.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "ctc_tc.h"
#include "kernels.hpp"

namespace ctc_internal {
int set_last_error(int code, const std::string &msg);  // api.cpp
}

namespace {

// Per-device staging for work tables: a small ring of (pinned host buffer, device buffer) pairs that
// only grow.  A slot's `done` event is recorded BEHIND the scoring kernel that reads the slot's device
// table, and the host waits on it before the slot is filled again, so calls on different streams (or
// host threads) never share a table that a kernel in flight still reads; up to kSlots calls per device
// are in flight before a call has to wait.
struct Slot {
    ctc::FastrWork *host = nullptr;
    ctc::FastrWork *dev = nullptr;
    size_t capacity = 0;
    hipEvent_t done = nullptr;  // the last kernel that read `dev`
};
constexpr int kSlots = 4;
struct Staging {
    Slot slot[kSlots];
    unsigned next = 0;
};
std::mutex g_mu;
std::map<int, Staging> g_staging;

}  // namespace

extern "C" int ctc_fastr_best_offsets(const uint8_t *d_ref, const uint8_t *d_read_seq, const uint8_t *d_quals,
                                      int read_stride, const int32_t *d_read_len,
                                      const int32_t *d_original_alignment, int n_regions,
                                      const int32_t *h_n_consensus, const int32_t *h_alt_reads,
                                      const int32_t *h_ref_lens, int32_t *d_best_score, int32_t *d_best_index,
                                      void *stream)
{
    using ctc_internal::set_last_error;
    try {  // the work table is host memory: no exception leaves the C ABI
    if (n_regions < 0 || read_stride <= 0) return set_last_error(CTC_TC_EINVAL, "bad n_regions or read_stride");
    if (n_regions == 0) return CTC_TC_OK;
    if (!d_ref || !d_read_seq || !d_quals || !d_read_len || !d_original_alignment || !h_n_consensus || !h_alt_reads ||
        !h_ref_lens || !d_best_score || !d_best_index)
        return set_last_error(CTC_TC_EINVAL, "null argument");

    const int per_block = ctc::fastr_reads_per_block();
    std::vector<ctc::FastrWork> work;
    int64_t ref_off = 0, read_off = 0, score_off = 0;
    int cons = 0;
    bool staged = read_stride <= ctc::fastr_max_read_lds();
    for (int t = 0; t < n_regions; ++t) {
        if (h_n_consensus[t] < 0 || h_alt_reads[t] < 0) return set_last_error(CTC_TC_EINVAL, "negative count");
        for (int k = 0; k < h_n_consensus[t]; ++k) {
            const int32_t len = h_ref_lens[cons];
            if (len < 0) return set_last_error(CTC_TC_EINVAL, "negative consensus length");
            if (len > ctc::fastr_max_ref_lds()) staged = false;
            for (int b = 0; b < h_alt_reads[t]; b += per_block) {
                ctc::FastrWork w;
                w.ref_off = ref_off;
                w.ref_len = len;
                w.n_reads = h_alt_reads[t] - b < per_block ? h_alt_reads[t] - b : per_block;
                w.first_read = read_off + b;
                w.score_off = score_off + b;
                work.push_back(w);
            }
            ref_off += len;
            cons += 1;
            score_off += h_alt_reads[t];
        }
        read_off += h_alt_reads[t];
    }
    if (work.empty()) return CTC_TC_OK;
    if (work.size() > 0x7FFFFFFFu) return set_last_error(CTC_TC_EUNSUPPORTED, "batch too large");

    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("hipGetDevice: ") + hipGetErrorString(e));
    hipStream_t s = static_cast<hipStream_t>(stream);

    std::lock_guard<std::mutex> lock(g_mu);
    Staging &ring = g_staging[device];
    Slot &st = ring.slot[ring.next++ % kSlots];
    // the upload and the kernel that used this slot last have finished with its two buffers
    if (st.done) (void)hipEventSynchronize(st.done);
    if (st.capacity < work.size()) {
        if (st.host) (void)hipHostFree(st.host);
        if (st.dev) (void)hipFree(st.dev);
        st.capacity = work.size() * 2;
        e = hipHostMalloc(reinterpret_cast<void **>(&st.host), st.capacity * sizeof(ctc::FastrWork), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&st.dev), st.capacity * sizeof(ctc::FastrWork));
        if (e != hipSuccess) {
            st.capacity = 0;
            return set_last_error(CTC_TC_ENOMEM, std::string("work table allocation: ") + hipGetErrorString(e));
        }
    }
    if (!st.done) {
        e = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
        if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
    memcpy(st.host, work.data(), work.size() * sizeof(ctc::FastrWork));
    e = hipMemcpyAsync(st.dev, st.host, work.size() * sizeof(ctc::FastrWork), hipMemcpyHostToDevice, s);
    if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("work table upload: ") + hipGetErrorString(e));

    ctc::FastrParams p;
    p.ref = d_ref;
    p.read_seq = d_read_seq;
    p.quals = d_quals;
    p.read_len = d_read_len;
    p.original_alignment = d_original_alignment;
    p.work = st.dev;
    p.best_score = d_best_score;
    p.best_index = d_best_index;
    p.read_stride = read_stride;
    e = ctc::launch_fastr(p, (int)work.size(), staged, s);
    (void)hipEventRecord(st.done, s);  // also after a failed launch: the upload still reads st.host
    if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("scoring kernel launch: ") + hipGetErrorString(e));
    return CTC_TC_OK;
    } catch (const std::bad_alloc &) {
        return set_last_error(CTC_TC_ENOMEM, "out of host memory for the work table");
    }
}

extern "C" size_t ctc_fastr_sw_workspace_bytes(int n_pairs, int max_ref_len, int max_read_len)
{
    if (n_pairs <= 0 || max_ref_len <= 0 || max_read_len <= 0) return 0;
    return (size_t)n_pairs * ctc::sw_bt_elems_per_pair(max_ref_len, max_read_len) * sizeof(short);
}

extern "C" int ctc_fastr_sw_align(const uint8_t *d_ref, const int64_t *d_ref_off, const int32_t *d_ref_len,
                                  const uint8_t *d_reads, int read_stride, const int32_t *d_read_len, int n_pairs,
                                  double w_match, double w_mismatch, double w_open, double w_extend, int max_ref_len,
                                  int max_read_len, int32_t *d_offset, int32_t *d_n_ops, uint8_t *d_op_type,
                                  int32_t *d_op_len, int max_ops, void *d_workspace, size_t workspace_bytes,
                                  void *stream)
{
    using ctc_internal::set_last_error;
    if (n_pairs < 0 || read_stride <= 0 || max_ops <= 0) return set_last_error(CTC_TC_EINVAL, "bad n_pairs, read_stride or max_ops");
    if (n_pairs == 0) return CTC_TC_OK;
    if (!d_ref || !d_ref_off || !d_ref_len || !d_reads || !d_read_len || !d_offset || !d_n_ops || !d_op_type || !d_op_len ||
        !d_workspace)
        return set_last_error(CTC_TC_EINVAL, "null argument");
    if (max_ref_len <= 0 || max_ref_len > 32767)
        return set_last_error(CTC_TC_EUNSUPPORTED, "max_ref_len must be in 1..32767 (back-track entries are int16)");
    if (max_read_len <= 0 || max_read_len > read_stride)
        return set_last_error(CTC_TC_EINVAL, "max_read_len must be in 1..read_stride");
    if (ctc::sw_lds_bytes(max_read_len, max_ops) > 64 * 1024)
        return set_last_error(CTC_TC_EUNSUPPORTED, "max_read_len and max_ops need more than 64 KB of LDS per workgroup");
    const size_t per_pair = ctc::sw_bt_elems_per_pair(max_ref_len, max_read_len) * sizeof(short);
    const size_t fit = workspace_bytes / per_pair;
    if (fit == 0) return set_last_error(CTC_TC_EINVAL, "workspace smaller than one pair's back-track store");
    ctc::SwParams p;
    p.ref = d_ref; p.reads = d_reads; p.ref_off = d_ref_off; p.ref_len = d_ref_len; p.read_len = d_read_len;
    p.read_stride = read_stride;
    p.w_match = w_match; p.w_mismatch = w_mismatch; p.w_open = w_open; p.w_extend = w_extend;
    p.max_ref_len = max_ref_len; p.max_read_len = max_read_len; p.max_ops = max_ops;
    p.bt = static_cast<short *>(d_workspace);
    p.bt_per_pair = per_pair / sizeof(short);
    p.offset = d_offset; p.n_ops = d_n_ops; p.op_len = d_op_len; p.op_type = d_op_type;
    // as many launches as the workspace asks for; they run one after the other on the stream
    for (long long first = 0; first < n_pairs; first += (long long)fit) {
        p.first_pair = first;
        p.n_pairs = (int)std::min<long long>((long long)fit, n_pairs - first);
        const hipError_t e = ctc::launch_sw(p, static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("alignment kernel launch: ") + hipGetErrorString(e));
    }
    return CTC_TC_OK;
}
