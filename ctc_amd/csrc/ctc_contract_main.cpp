// ctc_contract: the executable a CTC engine plugin names in its <command> template.
// The engine expands the template and runs it with system()
// (engine/src/ctc_plugin.cpp:166-206, engine/src/ctc_engine.cpp:155-161); this tool
// is a thin argv front-end over the C ABI in ctc_tc.h.  Tensors are raw
// little-endian dense files, packed with the last mode fastest.
//
//   ctc_contract --spec aeim,ebmj=abij --extents a=64,b=64,... --dtype f64
//                --a A.bin --b B.bin --out C.bin [--alpha 1 --beta 0]
//                [--gen-a SEED --gen-b SEED]   synthetic inputs instead of files
//                [--describe]                  print the plan and exit (needs no GPU)
//                [--shard MODE:RANK/NRANKS]    compute one slab only
//                [--ngpu N [--shard-on MODE]]  whole contraction on N GPUs of this node: one host thread
//                                              per GPU holding only its slabs; a free mode needs no exchange,
//                                              a contracted mode is summed by RCCL reduce-scatters
//                [--time ITERS]                print device ms per contraction
//   ctc_contract --permute abcd=dcba --extents a=..,b=.. --dtype f32 --a IN.bin --out OUT.bin
//                                              mode permutation of one tensor (input modes = output modes)
//
// Because the engine runs the command through a shell, write the spec with '='
// instead of '->' inside plugin templates.
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "ctc_tc.h"

namespace {

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "ctc_contract: %s\n", msg.c_str());
    exit(2);
}

void check(int rc, const char *what)
{
    if (rc < 0) die(std::string(what) + ": " + ctc_tc_last_error());
}

void hip_check(hipError_t e, const char *what)
{
    if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}

std::vector<char> read_file(const std::string &path, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) die("cannot open " + path);
    std::vector<char> buf(bytes);
    size_t got = fread(buf.data(), 1, bytes, f);
    int extra = fgetc(f);
    fclose(f);
    if (got != bytes || extra != EOF)
        die(path + ": size does not match the extents (" + std::to_string(bytes) + " bytes expected)");
    return buf;
}

void write_file(const std::string &path, const void *data, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) die("cannot create " + path);
    if (fwrite(data, 1, bytes, f) != bytes) die("short write to " + path);
    fclose(f);
}

}  // namespace

namespace {
int run(int argc, char **argv)
{
    std::map<std::string, std::string> opt;
    bool describe = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--describe") { describe = true; continue; }
        if (a == "--help" || a == "-h") {
            printf("usage: ctc_contract --spec S --extents a=..,b=.. [--dtype f64|f32] --a FILE --b FILE --out FILE\n"
                   "       [--alpha X] [--beta Y] [--gen-a SEED] [--gen-b SEED] [--describe]\n"
                   "       [--shard MODE:RANK/NRANKS] [--ngpu N [--shard-on MODE]] [--time ITERS] [--device N]\n"
                   "       ctc_contract --permute IN=OUT --extents ... [--dtype T] --a FILE --out FILE\n");
            return 0;
        }
        if (a.size() < 3 || a.substr(0, 2) != "--" || i + 1 >= argc) die("bad argument '" + a + "'");
        opt[a.substr(2)] = argv[++i];
    }
    if (!(opt.count("spec") || opt.count("permute")) || !opt.count("extents"))
        die("--spec (or --permute) and --extents are required");

    std::string names;
    std::vector<int64_t> extents;
    {
        const std::string &e = opt["extents"];
        size_t pos = 0;
        while (pos < e.size()) {
            size_t comma = e.find(',', pos);
            if (comma == std::string::npos) comma = e.size();
            std::string item = e.substr(pos, comma - pos);
            if (item.size() < 3 || item[1] != '=') die("bad --extents item '" + item + "'");
            names.push_back(item[0]);
            extents.push_back(atoll(item.c_str() + 2));
            pos = comma + 1;
        }
    }
    int dtype = CTC_TC_F64;
    if (opt.count("dtype")) {
        if (opt["dtype"] == "f64") dtype = CTC_TC_F64;
        else if (opt["dtype"] == "f32") dtype = CTC_TC_F32;
        else if (opt["dtype"] == "bf16") dtype = CTC_TC_BF16;
        else die("unknown --dtype " + opt["dtype"]);
    }
    const size_t es = dtype == CTC_TC_F64 ? 8 : dtype == CTC_TC_F32 ? 4 : 2;

    if (opt.count("permute")) {
        // IN=OUT (or IN->OUT): the same letters on both sides
        std::string ps = opt["permute"];
        size_t sep = ps.find("->"), seplen = 2;
        if (sep == std::string::npos) { sep = ps.find('='); seplen = 1; }
        if (sep == std::string::npos) die("bad --permute, want INMODES=OUTMODES");
        const std::string in_modes = ps.substr(0, sep), out_modes = ps.substr(sep + seplen);
        if (in_modes.empty() || in_modes.size() != out_modes.size()) die("--permute: both sides need the same modes");
        std::vector<int64_t> ext_in;
        std::vector<int> perm;
        size_t n = 1;
        for (char ch : in_modes) {
            size_t at = names.find(ch);
            if (at == std::string::npos) die(std::string("no extent for mode ") + ch);
            if (in_modes.find(ch) != in_modes.rfind(ch)) die(std::string("mode repeated in --permute: ") + ch);
            if (extents[at] < 1 || (size_t)extents[at] > (size_t(1) << 46) / n) die("--permute: bad extent or tensor too large");
            ext_in.push_back(extents[at]);
            n *= (size_t)extents[at];
        }
        for (char ch : out_modes) {
            size_t at = in_modes.find(ch);
            if (at == std::string::npos || out_modes.find(ch) != out_modes.rfind(ch))
                die("--permute: the output modes must be a permutation of the input modes");
            perm.push_back((int)at);
        }
        if (!opt.count("a") || !opt.count("out")) die("--permute needs --a IN and --out OUT");
        if (ctc_tc_device_count() < 1) die("no HIP device visible (this tool has no CPU path)");
        if (opt.count("device")) hip_check(hipSetDevice(atoi(opt["device"].c_str())), "hipSetDevice");
        std::vector<char> h = read_file(opt["a"], n * es);
        void *d_in = nullptr, *d_out = nullptr;
        hip_check(hipMalloc(&d_in, n * es), "hipMalloc in");
        hip_check(hipMalloc(&d_out, n * es), "hipMalloc out");
        hip_check(hipMemcpy(d_in, h.data(), n * es, hipMemcpyHostToDevice), "copy in");
        check(ctc_tc_permute(d_in, d_out, (int)perm.size(), ext_in.data(), perm.data(), (int)es, nullptr), "permute");
        hip_check(hipDeviceSynchronize(), "synchronize");
        hip_check(hipMemcpy(h.data(), d_out, n * es, hipMemcpyDeviceToHost), "copy out");
        write_file(opt["out"], h.data(), n * es);
        (void)hipFree(d_in);
        (void)hipFree(d_out);
        return 0;
    }

    ctc_tc_plan *plan = nullptr;
    check(ctc_tc_plan_create(opt["spec"].c_str(), (int)names.size(), names.data(), extents.data(), nullptr,
                             nullptr, nullptr, dtype, &plan), "plan");
    if (describe) {
        std::vector<char> buf(4096);
        check(ctc_tc_plan_describe(plan, buf.data(), buf.size()), "describe");
        fputs(buf.data(), stdout);
        return 0;
    }

    // operand sizes from the spec
    auto count = [&](const std::string &letters) {
        size_t n = 1;
        for (char ch : letters) {
            size_t at = names.find(ch);
            if (ch == ' ') continue;
            if (at == std::string::npos) die(std::string("no extent for mode ") + ch);
            n *= (size_t)extents[at];
        }
        return n;
    };
    std::string spec = opt["spec"];
    size_t comma = spec.find(',');
    size_t arrow = spec.find("->");
    size_t alen = 2;
    if (arrow == std::string::npos) { arrow = spec.find('='); alen = 1; }
    const size_t na = count(spec.substr(0, comma));
    const size_t nb = count(spec.substr(comma + 1, arrow - comma - 1));
    const size_t nc = count(spec.substr(arrow + alen));

    if (ctc_tc_device_count() < 1) die("no HIP device visible (this tool has no CPU path)");
    if (opt.count("device")) hip_check(hipSetDevice(atoi(opt["device"].c_str())), "hipSetDevice");
    if (!opt.count("out")) die("--out is required");
    const double alpha = opt.count("alpha") ? atof(opt["alpha"].c_str()) : 1.0;
    const double beta = opt.count("beta") ? atof(opt["beta"].c_str()) : 0.0;

    const int ngpu = opt.count("ngpu") ? atoi(opt["ngpu"].c_str()) : 1;
    if (ngpu < 1) die("bad --ngpu");
    if (ngpu > 1 || opt.count("shard-on")) {
        // One process, one host thread per GPU (the engine runs ONE command per plugin step).  Rank r holds
        // only its slabs: along a free (or batch) mode a slab of C and of the operand(s) that carry the mode,
        // no exchange, slabs of C merged on the host; along a CONTRACTED mode a slab of A and of B, the
        // partial results summed by RCCL reduce-scatters issued from the library (ctc_tc_sharded_execute),
        // after which rank r holds the flat range r of C.  Devices are shared round-robin when the node has
        // fewer than N; RCCL cannot put two ranks on one device, so in that case only (rehearsals on a
        // small box) the partial results are summed on the host.
        if (opt.count("shard")) die("--ngpu/--shard-on and --shard exclude each other");
        const std::string a_modes = spec.substr(0, comma), b_modes = spec.substr(comma + 1, arrow - comma - 1);
        const std::string c_modes = spec.substr(arrow + alen);
        auto has = [](const std::string &m, char ch) { return m.find(ch) != std::string::npos; };
        char mode = 0;
        if (opt.count("shard-on") && opt["shard-on"] != "auto") {
            mode = opt["shard-on"][0];
            if (!has(a_modes, mode) && !has(b_modes, mode)) die("--shard-on: not a mode of this contraction");
        } else {
            for (char ch : c_modes)  // the slowest free mode of C that divides evenly
                if (has(a_modes, ch) != has(b_modes, ch) && extents[names.find(ch)] % ngpu == 0) { mode = ch; break; }
            if (!mode) die("--ngpu: no free mode whose extent is a multiple of the GPU count");
        }
        const bool contracted = !has(c_modes, mode);
        const int64_t e_m = extents[names.find(mode)];
        if (e_m % ngpu != 0) die("--ngpu: the extent of the sharded mode must be a multiple of the GPU count");
        if (contracted && beta != 0.0) die("--beta with a contracted --shard-on is not supported (C would be added N times)");
        const int64_t cnt = e_m / ngpu;
        // geometry of the mode inside a packed operand: outer x e_m x inner
        struct Geo { bool present; size_t outer, inner; };
        auto geo = [&](const std::string &m) {
            Geo g{has(m, mode), 1, 1};
            if (!g.present) return g;
            const size_t at = m.find(mode);
            for (size_t i = 0; i < at; ++i) g.outer *= (size_t)extents[names.find(m[i])];
            for (size_t i = at + 1; i < m.size(); ++i) g.inner *= (size_t)extents[names.find(m[i])];
            return g;
        };
        const Geo ga = geo(a_modes), gb = geo(b_modes), gc = geo(c_modes);
        std::vector<char> hA, hB;
        if (!opt.count("gen-a")) { if (!opt.count("a")) die("--a or --gen-a is required"); hA = read_file(opt["a"], na * es); }
        if (!opt.count("gen-b")) { if (!opt.count("b")) die("--b or --gen-b is required"); hB = read_file(opt["b"], nb * es); }
        std::vector<char> hC(nc * es, 0);
        if (beta != 0.0) hC = read_file(opt["out"], nc * es);
        const int ndev = ctc_tc_device_count();
        const bool use_rccl = contracted && ndev >= ngpu;
        std::vector<void *> comms((size_t)ngpu, nullptr);
        if (use_rccl) check(ctc_tc_comm_init_all(comms.data(), ngpu, nullptr), "communicators");
        std::vector<int64_t> local_ext = extents;
        local_ext[names.find(mode)] = cnt;
        std::vector<std::vector<char>> partials;  // host-side sum only
        if (contracted && !use_rccl) partials.resize((size_t)ngpu);
        std::vector<std::string> errors((size_t)ngpu);
        std::vector<double> rank_ms((size_t)ngpu, 0.0);
        // Collectives need every rank: a rank that fails while setting up must not leave its peers waiting
        // inside one for ever.  Every worker finishes its fallible set-up (allocations, uploads, plans, tables),
        // then all meet here; the collectives are entered only if every rank got this far.  A rank that fails
        // later, between collectives, aborts every communicator so that its peers' calls return with an error.
        std::mutex gate_mu;
        std::condition_variable gate_cv;
        int gate_arrived = 0;
        bool gate_all_ok = true;
        auto meet = [&](bool ok) {
            std::unique_lock<std::mutex> lock(gate_mu);
            gate_all_ok = gate_all_ok && ok;
            if (++gate_arrived == ngpu) gate_cv.notify_all();
            else gate_cv.wait(lock, [&] { return gate_arrived == ngpu; });
            return gate_all_ok;
        };
        std::atomic<bool> aborted{false};
        auto abort_collectives = [&]() {
            if (use_rccl && !aborted.exchange(true))
                for (void *c : comms) (void)ctc_tc_comm_abort(c);  // frees them as well
        };
        auto worker = [&](int r) {
            bool met = !contracted;  // only the contracted path has a meeting point
            auto fail = [&](const std::string &m) {
                errors[(size_t)r] = m;
                if (!met) {
                    met = true;
                    (void)meet(false);
                } else if (contracted) {
                    abort_collectives();
                }
            };
            if (hipSetDevice(r % ndev) != hipSuccess) return fail("hipSetDevice");
            // for the tests of the failure path: CTC_CONTRACT_FAIL_RANK=r makes rank r report a set-up failure
            if (getenv("CTC_CONTRACT_FAIL_RANK") && atoi(getenv("CTC_CONTRACT_FAIL_RANK")) == r) return fail("set-up failure requested by CTC_CONTRACT_FAIL_RANK");
            // slab [r cnt, (r+1) cnt) of `mode` of a packed operand -> packed device buffer
            auto upload = [&](const std::vector<char> &host, const char *gen, size_t n_full, const Geo &g, void **dev) -> bool {
                const size_t n_loc = g.present ? n_full / (size_t)ngpu : n_full;
                if (hipMalloc(dev, n_loc * es) != hipSuccess) return false;
                const size_t width = (size_t)cnt * g.inner * es, pitch = (size_t)e_m * g.inner * es;
                if (host.empty()) {
                    // synthetic operand: the slab of the full stream, cut out on the device
                    if (!g.present) return ctc_tc_fill(*dev, (int64_t)n_full, strtoull(gen, nullptr, 0), 0, dtype, nullptr) >= 0;
                    void *full = nullptr;
                    if (hipMalloc(&full, n_full * es) != hipSuccess) return false;
                    bool ok = ctc_tc_fill(full, (int64_t)n_full, strtoull(gen, nullptr, 0), 0, dtype, nullptr) >= 0;
                    ok = ok && hipMemcpy2D(*dev, width, static_cast<char *>(full) + (size_t)r * width, pitch, width, g.outer,
                                           hipMemcpyDeviceToDevice) == hipSuccess;
                    (void)hipFree(full);
                    return ok;
                }
                if (!g.present) return hipMemcpy(*dev, host.data(), n_full * es, hipMemcpyHostToDevice) == hipSuccess;
                return hipMemcpy2D(*dev, width, host.data() + (size_t)r * width, pitch, width, g.outer, hipMemcpyHostToDevice) ==
                       hipSuccess;
            };
            void *dA = nullptr, *dB = nullptr, *dC = nullptr, *dW = nullptr;
            if (!upload(hA, opt.count("gen-a") ? opt["gen-a"].c_str() : "0", na, ga, &dA)) return fail("operand A upload");
            if (!upload(hB, opt.count("gen-b") ? opt["gen-b"].c_str() : "0", nb, gb, &dB)) return fail("operand B upload");
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail("hipEventCreate");
            const int iters = opt.count("time") ? atoi(opt["time"].c_str()) : 0;
            if (contracted) {
                ctc_tc_sharded *sh = nullptr;
                const int reduce = use_rccl ? CTC_TC_REDUCE_SCATTER : CTC_TC_ALL_REDUCE;  // all-reduce form: whole partial C, no workspace
                if (ctc_tc_sharded_create(spec.c_str(), (int)names.size(), names.data(), extents.data(), dtype, mode, r, ngpu, 4,
                                          reduce, &sh) < 0)
                    return fail(std::string("sharded plan: ") + ctc_tc_last_error());
                const size_t n_out = (size_t)ctc_tc_sharded_get(sh, CTC_TC_SQ_OUT_ELEMS, 0);
                const size_t n_ws = (size_t)ctc_tc_sharded_get(sh, CTC_TC_SQ_WORKSPACE_ELEMS, 0);
                if (hipMalloc(&dC, n_out * es) != hipSuccess || (n_ws && hipMalloc(&dW, n_ws * es) != hipSuccess)) return fail("hipMalloc");
                if (ctc_tc_sharded_prepare(sh) < 0) return fail(std::string("prepare: ") + ctc_tc_last_error());
                met = true;
                if (!meet(true)) {
                    errors[(size_t)r] = "not started: another rank failed during set-up";
                    return;
                }
                for (int it = 0; it < iters + 1; ++it) {  // the last execution is the one whose result is kept
                    if (it == 0 && iters > 0) (void)hipEventRecord(e0, nullptr);
                    if (it == iters && iters > 0) (void)hipEventRecord(e1, nullptr);
                    // without RCCL (ranks share a device) no communicator is passed: the call multiplies only
                    if (ctc_tc_sharded_execute(sh, use_rccl ? comms[(size_t)r] : nullptr, alpha, dA, dB, dC, dW, (int64_t)(n_ws * es),
                                               nullptr) < 0)
                        return fail(std::string("execute: ") + ctc_tc_last_error());
                }
                if (hipDeviceSynchronize() != hipSuccess) return fail("synchronize");
                if (use_rccl) {
                    if (hipMemcpy(hC.data() + (size_t)r * n_out * es, dC, n_out * es, hipMemcpyDeviceToHost) != hipSuccess)
                        return fail("copy C back");
                } else {
                    partials[(size_t)r].resize(n_out * es);
                    if (hipMemcpy(partials[(size_t)r].data(), dC, n_out * es, hipMemcpyDeviceToHost) != hipSuccess) return fail("copy C back");
                }
                ctc_tc_sharded_destroy(sh);
            } else {
                ctc_tc_plan *slab = nullptr;
                if (ctc_tc_plan_create(spec.c_str(), (int)names.size(), names.data(), local_ext.data(), nullptr, nullptr, nullptr, dtype,
                                       &slab) < 0)
                    return fail(std::string("slab plan: ") + ctc_tc_last_error());
                const size_t n_loc = nc / (size_t)ngpu;
                const size_t width = (size_t)cnt * gc.inner * es, pitch = (size_t)e_m * gc.inner * es;
                if (hipMalloc(&dC, n_loc * es) != hipSuccess) return fail("hipMalloc");
                auto upload_c = [&]() {
                    return beta == 0.0 || hipMemcpy2D(dC, width, hC.data() + (size_t)r * width, pitch, width, gc.outer,
                                                      hipMemcpyHostToDevice) == hipSuccess;
                };
                if (!upload_c()) return fail("C upload");
                if (iters > 0) {
                    double ms = 0;
                    if (ctc_tc_time_execute(slab, alpha, dA, dB, beta, dC, nullptr, 2, iters, &ms) < 0)
                        return fail(std::string("time: ") + ctc_tc_last_error());
                    rank_ms[(size_t)r] = ms;
                    if (!upload_c()) return fail("C upload");  // timing executed beta * C over and over
                }
                if (ctc_tc_execute(slab, alpha, dA, dB, beta, dC, nullptr) < 0) return fail(std::string("execute: ") + ctc_tc_last_error());
                if (hipDeviceSynchronize() != hipSuccess) return fail("synchronize");
                // ranks write disjoint runs of the host image
                if (hipMemcpy2D(hC.data() + (size_t)r * width, pitch, dC, width, width, gc.outer, hipMemcpyDeviceToHost) != hipSuccess)
                    return fail("copy C back");
                ctc_tc_plan_destroy(slab);
            }
            if (contracted && iters > 0) {
                float ms = 0;
                (void)hipEventElapsedTime(&ms, e0, e1);
                rank_ms[(size_t)r] = (double)ms / iters;
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            (void)hipFree(dA);
            (void)hipFree(dB);
            (void)hipFree(dC);
            if (dW) (void)hipFree(dW);
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < ngpu; ++r) threads.emplace_back(worker, r);
        for (auto &th : threads) th.join();
        {
            // report the rank that failed by itself before the ranks that merely did not start
            std::string first;
            for (int r = 0; r < ngpu; ++r)
                if (!errors[(size_t)r].empty() && (first.empty() || first.find("not started") != std::string::npos))
                    first = "rank " + std::to_string(r) + ": " + errors[(size_t)r];
            if (!first.empty()) {
                abort_collectives();
                die(first);
            }
        }
        if (!aborted.load())
            for (void *c : comms) ctc_tc_comm_destroy(c);
        if (contracted && !use_rccl) {
            // ranks shared devices: sum the partial results here, in rank order
            auto sum = [&](auto *dst, size_t n) {
                typedef typename std::remove_pointer<decltype(dst)>::type T;
                for (int r = 0; r < ngpu; ++r) {
                    const T *src = reinterpret_cast<const T *>(partials[(size_t)r].data());
                    for (size_t i = 0; i < n; ++i) dst[i] = r == 0 ? src[i] : dst[i] + src[i];
                }
            };
            if (dtype == CTC_TC_F64) sum(reinterpret_cast<double *>(hC.data()), nc);
            else if (dtype == CTC_TC_F32) sum(reinterpret_cast<float *>(hC.data()), nc);
            else die("summing bf16 partial results on the host is not supported: use as many GPUs as ranks");
        }
        if (opt.count("time")) {
            double worst = 0;
            for (double ms : rank_ms) worst = ms > worst ? ms : worst;
            int64_t flops = 0;
            ctc_tc_plan_get(plan, CTC_TC_Q_FLOPS, &flops, 1);
            printf("ngpu %d mode %c %s devices %d max_device_ms %.4f tflops %.3f\n", ngpu, mode,
                   contracted ? (use_rccl ? "contracted:rccl_reduce_scatter" : "contracted:host_sum") : "free", ndev < ngpu ? ndev : ngpu,
                   worst, (double)flops / worst * 1e-9);
        }
        write_file(opt["out"], hC.data(), nc * es);
        ctc_tc_plan_destroy(plan);
        return 0;
    }

    void *dA = nullptr, *dB = nullptr, *dC = nullptr;
    hip_check(hipMalloc(&dA, na * es), "hipMalloc A");
    hip_check(hipMalloc(&dB, nb * es), "hipMalloc B");
    hip_check(hipMalloc(&dC, nc * es), "hipMalloc C");
    if (opt.count("gen-a")) {
        check(ctc_tc_fill(dA, (int64_t)na, strtoull(opt["gen-a"].c_str(), nullptr, 0), 0, dtype, nullptr), "fill A");
    } else {
        if (!opt.count("a")) die("--a or --gen-a is required");
        std::vector<char> h = read_file(opt["a"], na * es);
        hip_check(hipMemcpy(dA, h.data(), na * es, hipMemcpyHostToDevice), "copy A");
    }
    if (opt.count("gen-b")) {
        check(ctc_tc_fill(dB, (int64_t)nb, strtoull(opt["gen-b"].c_str(), nullptr, 0), 0, dtype, nullptr), "fill B");
    } else {
        if (!opt.count("b")) die("--b or --gen-b is required");
        std::vector<char> h = read_file(opt["b"], nb * es);
        hip_check(hipMemcpy(dB, h.data(), nb * es, hipMemcpyHostToDevice), "copy B");
    }
    std::vector<char> hC(nc * es, 0);
    if (beta != 0.0) {
        hC = read_file(opt["out"], nc * es);  // C is read-modify-write when beta != 0
    }
    hip_check(hipMemcpy(dC, hC.data(), nc * es, hipMemcpyHostToDevice), "copy C");

    ctc_tc_plan *run = plan;
    int64_t oa = 0, ob = 0, oc = 0;
    if (opt.count("shard")) {
        char mode = 0;
        int rank = 0, nranks = 1;
        if (sscanf(opt["shard"].c_str(), "%c:%d/%d", &mode, &rank, &nranks) != 3) die("bad --shard, want MODE:RANK/NRANKS");
        check(ctc_tc_plan_shard(plan, mode, rank, nranks, &run, &oa, &ob, &oc), "shard");
    }
    const void *pa = static_cast<char *>(dA) + oa * es;
    const void *pb = static_cast<char *>(dB) + ob * es;
    void *pc = static_cast<char *>(dC) + oc * es;
    if (opt.count("time")) {
        double ms = 0;
        check(ctc_tc_time_execute(run, alpha, pa, pb, beta, pc, nullptr, 2, atoi(opt["time"].c_str()), &ms), "time");
        int64_t flops = 0;
        ctc_tc_plan_get(run, CTC_TC_Q_FLOPS, &flops, 1);
        printf("device_ms %.4f tflops %.3f\n", ms, (double)flops / ms * 1e-9);
        // timing executed beta * C over and over: start the kept execution from the caller's C again
        if (beta != 0.0) hip_check(hipMemcpy(dC, hC.data(), nc * es, hipMemcpyHostToDevice), "copy C");
    }
    check(ctc_tc_execute(run, alpha, pa, pb, beta, pc, nullptr), "execute");
    hip_check(hipDeviceSynchronize(), "synchronize");
    hip_check(hipMemcpy(hC.data(), dC, nc * es, hipMemcpyDeviceToHost), "copy C back");
    write_file(opt["out"], hC.data(), nc * es);
    if (run != plan) ctc_tc_plan_destroy(run);
    ctc_tc_plan_destroy(plan);
    (void)hipFree(dA);
    (void)hipFree(dB);
    (void)hipFree(dC);
    return 0;
}
}  // namespace

int main(int argc, char **argv)
{
    // a size the host cannot hold (extents that do not describe the files) ends as a message, not as an abort
    try {
        return run(argc, argv);
    } catch (const std::bad_alloc &) {
        die("out of host memory (do the extents describe the tensors?)");
    } catch (const std::exception &e) {
        die(std::string("internal error: ") + e.what());
    }
}
