// Contraction sharded along a CONTRACTED mode over the GPUs of a node (include/ctc_tc.h,
// ctc_tc_sharded_*): every rank multiplies its 1/p of A and B into a partial C and the partials
// are summed by RCCL.  The sum is pipelined against the GEMM: C is cut into chunks along its
// slowest mode, chunk i is reduced on a second stream while chunk i + 1 is multiplied.
//
// Reduce-scatter with contiguous ownership.  Rank r is to end up with the flat range
// [r |C|/p, (r+1) |C|/p) of C.  A chunk therefore cannot be a contiguous piece of C (its
// reduce-scatter would hand rank r a piece of every chunk, but of the wrong rows): chunk c is the
// union over r of the c-th sub-range of rank r's range.  With s = C's slowest mode (extent E, a free
// or batch mode), that is s = (s1, s2) with s1 = p outer blocks and chunk c = all s1, s2 in
// [c E/(p n), (c+1) E/(p n)).  The chunk's GEMM is ONE launch: the planner takes the split mode with
// explicit strides, the partial result goes to a staging buffer laid out [s1][s2 in chunk][rest],
// which is exactly the send buffer ncclReduceScatter wants, and rank r receives its piece at
// offset c |C|/(p n) of its output.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "api_internal.hpp"

using ctc_internal::set_last_error;
using ctc_internal::set_last_hip_error;

namespace {

int fail_nccl(ncclResult_t r, const char *what)
{
    return set_last_error(CTC_TC_EDEVICE, std::string(what) + ": " + ncclGetErrorString(r));
}

struct Chunk {
    std::unique_ptr<ctc_tc_plan> plan;
    int64_t off_a = 0, off_b = 0;  // element offsets into the rank-local operands
    int64_t off_part = 0;          // into the partial-result buffer (reduce-scatter) or into C (all-reduce)
    int64_t off_out = 0;           // into the rank's output (reduce-scatter)
    int64_t elems = 0;             // elements of C this chunk covers (all ranks' pieces together)
};

}  // namespace

struct ctc_tc_sharded {
    int rank = 0, nranks = 1, reduce = 0, dtype = 0;
    int64_t c_elems = 0;  // |C|
    std::vector<Chunk> chunks;
    // per-object stream and events of the device it first ran on
    int device = -1;
    hipStream_t comm_stream = nullptr;
    std::vector<hipEvent_t> chunk_done;
    hipEvent_t all_done = nullptr;
};

namespace {

ncclDataType_t nccl_type(int dtype)
{
    return dtype == CTC_TC_F64 ? ncclDouble : (dtype == CTC_TC_F32 ? ncclFloat : ncclBfloat16);
}

int ensure_streams(ctc_tc_sharded *s)
{
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return set_last_hip_error(e, "hipGetDevice");
    if (s->device == device) return CTC_TC_OK;
    if (s->device != -1) return set_last_error(CTC_TC_EINVAL, "a sharded contraction stays on the device it first ran on");
    e = hipStreamCreateWithFlags(&s->comm_stream, hipStreamNonBlocking);
    if (e != hipSuccess) return set_last_hip_error(e, "hipStreamCreate");
    s->chunk_done.resize(s->chunks.size());
    for (auto &ev : s->chunk_done) {
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return set_last_hip_error(e, "hipEventCreate");
    }
    e = hipEventCreateWithFlags(&s->all_done, hipEventDisableTiming);
    if (e != hipSuccess) return set_last_hip_error(e, "hipEventCreate");
    s->device = device;
    return CTC_TC_OK;
}

}  // namespace

extern "C" {

int ctc_tc_comm_unique_id(void *id128)
{
    static_assert(sizeof(ncclUniqueId) == CTC_TC_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id128) return set_last_error(CTC_TC_EINVAL, "null argument");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail_nccl(r, "ncclGetUniqueId");
    memcpy(id128, &id, sizeof(id));
    return CTC_TC_OK;
}

int ctc_tc_comm_init_rank(void **comm, int nranks, int rank, const void *id128)
{
    if (!comm || !id128 || nranks < 1 || rank < 0 || rank >= nranks) return set_last_error(CTC_TC_EINVAL, "bad argument");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);
    if (r != ncclSuccess) return fail_nccl(r, "ncclCommInitRank");
    *comm = c;
    return CTC_TC_OK;
}

int ctc_tc_comm_init_all(void **comms, int ndev, const int *devices)
{
    if (!comms || ndev < 1) return set_last_error(CTC_TC_EINVAL, "bad argument");
    std::vector<ncclComm_t> c((size_t)ndev, nullptr);
    ncclResult_t r = ncclCommInitAll(c.data(), ndev, devices);
    if (r != ncclSuccess) return fail_nccl(r, "ncclCommInitAll");
    for (int i = 0; i < ndev; ++i) comms[i] = c[(size_t)i];
    return CTC_TC_OK;
}

int ctc_tc_comm_destroy(void *comm)
{
    if (!comm) return CTC_TC_OK;
    ncclResult_t r = ncclCommDestroy(static_cast<ncclComm_t>(comm));
    if (r != ncclSuccess) return fail_nccl(r, "ncclCommDestroy");
    return CTC_TC_OK;
}

int ctc_tc_comm_abort(void *comm)
{
    if (!comm) return CTC_TC_OK;
    ncclResult_t r = ncclCommAbort(static_cast<ncclComm_t>(comm));
    if (r != ncclSuccess) return fail_nccl(r, "ncclCommAbort");
    return CTC_TC_OK;
}

int ctc_tc_sharded_create(const char *spec, int n_ext, const char *mode_names, const int64_t *extents, int dtype,
                          char mode, int rank, int nranks, int nchunks, int reduce, ctc_tc_sharded **out)
{
    CTC_ABI_TRY
    if (!spec || !mode_names || !extents || !out || n_ext <= 0) return set_last_error(CTC_TC_EINVAL, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return set_last_error(CTC_TC_EINVAL, "bad rank/nranks");
    if (reduce != CTC_TC_REDUCE_SCATTER && reduce != CTC_TC_ALL_REDUCE) return set_last_error(CTC_TC_EINVAL, "bad reduce kind");

    // the rank-local contraction: the sharded mode shrinks to extent / p
    std::string names(mode_names, (size_t)n_ext);
    std::vector<int64_t> ext(extents, extents + n_ext);
    const size_t at = names.find(mode);
    if (at == std::string::npos) return set_last_error(CTC_TC_EINVAL, std::string("unknown mode '") + mode + "'");
    if (ext[at] % nranks) return set_last_error(CTC_TC_EINVAL, "extent of the sharded mode must be a multiple of the rank count");
    ext[at] /= nranks;
    ctc::Plan local;
    std::string err = ctc::build_plan(spec, n_ext, names.data(), ext.data(), nullptr, nullptr, nullptr, dtype, local);
    if (!err.empty()) return set_last_error(CTC_TC_ESPEC, err);
    int mode_id = -1;
    for (size_t i = 0; i < local.modes.size(); ++i)
        if (local.modes[i].name == mode) mode_id = (int)i;
    if (mode_id < 0 || local.modes[(size_t)mode_id].cls != ctc::kContracted)
        return set_last_error(CTC_TC_EINVAL, "ctc_tc_sharded_* shards a CONTRACTED mode (a free or batch mode needs no exchange: "
                                             "ctc_tc_plan_shard)");
    if (local.c_modes.empty()) return set_last_error(CTC_TC_EUNSUPPORTED, "scalar result");

    std::unique_ptr<ctc_tc_sharded> s(new ctc_tc_sharded());
    s->rank = rank; s->nranks = nranks; s->reduce = reduce; s->dtype = dtype;
    s->c_elems = local.size_c;
    if (reduce == CTC_TC_REDUCE_SCATTER && s->c_elems % nranks)
        return set_last_error(CTC_TC_EINVAL, "reduce-scatter needs |C| divisible by the rank count");

    // chunking along C's slowest mode
    const ctc::Mode &slow = local.modes[(size_t)local.c_modes[0]];
    const int64_t E = slow.extent;
    const int64_t blocks = reduce == CTC_TC_REDUCE_SCATTER ? nranks : 1;  // s1
    if (nchunks < 1) nchunks = 1;
    bool split = (E % blocks == 0);
    int64_t n = nchunks;
    if (split) {
        const int64_t per_block = E / blocks;
        while (n > 1 && per_block % n) --n;  // the largest chunk count <= nchunks that divides evenly
    } else {
        n = 1;  // ownership boundaries cut through rows of the slowest mode: one plain reduce-scatter of everything
    }
    // packed local strides of A, B, C and the position of the slow mode in each
    auto packed = [&](const std::vector<int> &ids) {
        std::vector<int64_t> st(ids.size(), 1);
        for (int i = (int)ids.size() - 2; i >= 0; --i) st[(size_t)i] = st[(size_t)i + 1] * local.modes[(size_t)ids[(size_t)i + 1]].extent;
        return st;
    };
    const std::vector<int64_t> st_a = packed(local.a_modes), st_b = packed(local.b_modes), st_c = packed(local.c_modes);
    auto find_pos = [&](const std::vector<int> &ids) {
        for (size_t i = 0; i < ids.size(); ++i)
            if (ids[i] == local.c_modes[0]) return (int)i;
        return -1;
    };
    const int pos_a = find_pos(local.a_modes), pos_b = find_pos(local.b_modes);
    // a letter for the outer part of the split mode
    char outer = 0;
    for (char ch = 'A'; ch <= 'z' && !outer; ++ch) {
        const bool letter = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z');
        if (letter && names.find(ch) == std::string::npos) outer = ch;
    }
    const bool two_level = split && blocks > 1;
    if (two_level && (!outer || (int)local.modes.size() + 1 > ctc::kMaxModes))
        return set_last_error(CTC_TC_EUNSUPPORTED, "no spare mode to split C's slowest mode with");

    const int64_t rows_per_chunk = two_level ? E / blocks / n : E / n;  // extent of the (inner) slow mode in one chunk
    const int64_t row_elems = st_c[0];                                   // elements of C per index of its slowest mode
    for (int64_t c = 0; c < n; ++c) {
        Chunk ch;
        ch.plan.reset(new ctc_tc_plan());
        // spec, extents and strides of this chunk
        std::string sa, sb, sc;
        std::vector<int64_t> xa, xb;  // strides
        auto build_operand = [&](const std::vector<int> &ids, const std::vector<int64_t> &st, std::string &letters,
                                 std::vector<int64_t> &strides) {
            for (size_t i = 0; i < ids.size(); ++i) {
                if (two_level && ids[i] == local.c_modes[0]) {
                    letters.push_back(outer);
                    strides.push_back(st[i] * (E / blocks));
                }
                letters.push_back(local.modes[(size_t)ids[i]].name);
                strides.push_back(st[i]);
            }
        };
        build_operand(local.a_modes, st_a, sa, xa);
        build_operand(local.b_modes, st_b, sb, xb);
        for (size_t i = 0; i < local.c_modes.size(); ++i) {
            if (two_level && i == 0) sc.push_back(outer);
            sc.push_back(local.modes[(size_t)local.c_modes[i]].name);
        }
        std::string cn;
        std::vector<int64_t> ce;
        for (const ctc::Mode &m : local.modes) {
            cn.push_back(m.name);
            ce.push_back(m.name == slow.name ? rows_per_chunk : m.extent);
        }
        if (two_level) {
            cn.push_back(outer);
            ce.push_back(blocks);
        }
        // C of the chunk is packed in [outer][slow in chunk][rest] order: the staging piece (reduce-scatter)
        // or the chunk's rows of C itself (all-reduce, where outer does not exist)
        err = ctc::build_plan(sa + "," + sb + "->" + sc, (int)cn.size(), cn.data(), ce.data(), xa.data(), xb.data(), nullptr,
                              dtype, ch.plan->plan);
        if (!err.empty()) return set_last_error(CTC_TC_ESPEC, "chunk plan: " + err);
        const int64_t first_row = c * rows_per_chunk;
        ch.off_a = pos_a >= 0 ? first_row * st_a[(size_t)pos_a] : 0;
        ch.off_b = pos_b >= 0 ? first_row * st_b[(size_t)pos_b] : 0;
        ch.elems = (two_level ? blocks : 1) * rows_per_chunk * row_elems;
        ch.off_part = c * ch.elems;
        ch.off_out = reduce == CTC_TC_REDUCE_SCATTER ? c * (ch.elems / nranks) : 0;
        s->chunks.push_back(std::move(ch));
    }
    *out = s.release();
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

int ctc_tc_sharded_destroy(ctc_tc_sharded *s)
{
    if (!s) return CTC_TC_OK;
    for (Chunk &c : s->chunks) {
        ctc_tc_plan *p = c.plan.release();
        ctc_tc_plan_destroy(p);
    }
    for (hipEvent_t ev : s->chunk_done)
        if (ev) (void)hipEventDestroy(ev);
    if (s->all_done) (void)hipEventDestroy(s->all_done);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    delete s;
    return CTC_TC_OK;
}

int64_t ctc_tc_sharded_get(const ctc_tc_sharded *s, int query, int64_t chunk)
{
    if (!s) return set_last_error(CTC_TC_EINVAL, "null argument");
    const bool per_chunk = query >= CTC_TC_SQ_CHUNK_OFF_A;
    if (per_chunk && (chunk < 0 || chunk >= (int64_t)s->chunks.size())) return set_last_error(CTC_TC_EINVAL, "bad chunk index");
    switch (query) {
    case CTC_TC_SQ_CHUNKS: return (int64_t)s->chunks.size();
    case CTC_TC_SQ_C_ELEMS: return s->c_elems;
    case CTC_TC_SQ_OUT_ELEMS: return s->reduce == CTC_TC_REDUCE_SCATTER ? s->c_elems / s->nranks : s->c_elems;
    case CTC_TC_SQ_WORKSPACE_ELEMS: return s->reduce == CTC_TC_REDUCE_SCATTER ? s->c_elems : 0;
    case CTC_TC_SQ_CHUNK_OFF_A: return s->chunks[(size_t)chunk].off_a;
    case CTC_TC_SQ_CHUNK_OFF_B: return s->chunks[(size_t)chunk].off_b;
    case CTC_TC_SQ_CHUNK_OFF_PARTIAL: return s->chunks[(size_t)chunk].off_part;
    case CTC_TC_SQ_CHUNK_OFF_OUT: return s->chunks[(size_t)chunk].off_out;
    case CTC_TC_SQ_CHUNK_ELEMS: return s->chunks[(size_t)chunk].elems;
    default: return set_last_error(CTC_TC_EINVAL, "unknown query");
    }
}

ctc_tc_plan *ctc_tc_sharded_chunk_plan(const ctc_tc_sharded *s, int chunk)
{
    if (!s || chunk < 0 || chunk >= (int)s->chunks.size()) {
        set_last_error(CTC_TC_EINVAL, "bad chunk index");
        return nullptr;
    }
    return s->chunks[(size_t)chunk].plan.get();
}

int ctc_tc_sharded_prepare(ctc_tc_sharded *s)
{
    CTC_ABI_TRY
    if (!s) return set_last_error(CTC_TC_EINVAL, "null argument");
    int rc = ensure_streams(s);
    for (size_t c = 0; c < s->chunks.size() && rc == CTC_TC_OK; ++c) rc = ctc_tc_plan_prepare(s->chunks[c].plan.get());
    return rc;
    CTC_ABI_CATCH
}

int ctc_tc_sharded_execute(ctc_tc_sharded *s, void *nccl_comm, double alpha, const void *A, const void *B, void *C,
                           void *workspace, int64_t workspace_bytes, void *stream)
{
    CTC_ABI_TRY
    if (!s || !A || !B || !C) return set_last_error(CTC_TC_EINVAL, "null argument");
    if (!nccl_comm && s->nranks > 1 && s->reduce == CTC_TC_REDUCE_SCATTER)
        return set_last_error(CTC_TC_EINVAL, "null communicator (without one only the all-reduce form runs: it then leaves "
                                             "this rank's partial C for the caller to sum)");
    const int64_t es = ctc::dtype_size(s->dtype);
    const bool scatter = s->reduce == CTC_TC_REDUCE_SCATTER;
    if (scatter && (!workspace || workspace_bytes < s->c_elems * es))
        return set_last_error(CTC_TC_EINVAL, "reduce-scatter needs a workspace of |C| elements for the partial result");
    int rc = ensure_streams(s);
    if (rc) return rc;
    hipStream_t compute = static_cast<hipStream_t>(stream);
    ncclComm_t comm = static_cast<ncclComm_t>(nccl_comm);
    const char *a = static_cast<const char *>(A), *b = static_cast<const char *>(B);
    char *part = scatter ? static_cast<char *>(workspace) : static_cast<char *>(C);
    char *outp = static_cast<char *>(C);
    for (size_t c = 0; c < s->chunks.size(); ++c) {
        Chunk &ch = s->chunks[c];
        rc = ctc_internal::launch_plan(ch.plan.get(), alpha, a + ch.off_a * es, b + ch.off_b * es, 0.0,
                                       part + ch.off_part * es, compute);
        if (rc) return rc;
        hipError_t e = hipEventRecord(s->chunk_done[c], compute);
        if (e == hipSuccess) e = hipStreamWaitEvent(s->comm_stream, s->chunk_done[c], 0);
        if (e != hipSuccess) return set_last_hip_error(e, "chunk hand-over to the communication stream");
        ncclResult_t r = ncclSuccess;
        if (comm) {
            if (scatter)
                r = ncclReduceScatter(part + ch.off_part * es, outp + ch.off_out * es, (size_t)(ch.elems / s->nranks),
                                      nccl_type(s->dtype), ncclSum, comm, s->comm_stream);
            else
                r = ncclAllReduce(part + ch.off_part * es, part + ch.off_part * es, (size_t)ch.elems, nccl_type(s->dtype),
                                  ncclSum, comm, s->comm_stream);
        } else if (scatter) {
            // one rank and no communicator: the reduce-scatter is a copy
            e = hipMemcpyAsync(outp + ch.off_out * es, part + ch.off_part * es, (size_t)(ch.elems * es), hipMemcpyDeviceToDevice,
                               s->comm_stream);
            if (e != hipSuccess) return set_last_hip_error(e, "hipMemcpyAsync");
        }
        if (r != ncclSuccess) return fail_nccl(r, scatter ? "ncclReduceScatter" : "ncclAllReduce");
    }
    // the caller's stream continues when the last collective is done (and the partial buffer is free again)
    hipError_t e = hipEventRecord(s->all_done, s->comm_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(compute, s->all_done, 0);
    if (e != hipSuccess) return set_last_hip_error(e, "joining the communication stream");
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

}  // extern "C"
