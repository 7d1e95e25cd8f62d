// C ABI of the FASTR scoring path (include/ctc_fastr.h): builds the work table from the
// per-region counts on the host, uploads it through a pinned staging buffer and launches.
#include "ctc_fastr.h"

#include <hip/hip_runtime.h>

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "ctc_tc.h"
#include "kernels.hpp"

namespace ctc_internal {
int set_last_error(int code, const std::string &msg);  // api.cpp
}

namespace {

// per-device staging for work tables: a pinned host ring and a device buffer that only grow
struct Staging {
    ctc::FastrWork *host = nullptr;
    ctc::FastrWork *dev = nullptr;
    size_t capacity = 0;
    hipEvent_t done = nullptr;  // last upload that read `host`
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
    Staging &st = g_staging[device];
    if (st.done) (void)hipEventSynchronize(st.done);  // the previous upload has consumed the pinned buffer
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
    (void)hipEventRecord(st.done, s);

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
    if (e != hipSuccess) return set_last_error(CTC_TC_EDEVICE, std::string("scoring kernel launch: ") + hipGetErrorString(e));
    return CTC_TC_OK;
}
