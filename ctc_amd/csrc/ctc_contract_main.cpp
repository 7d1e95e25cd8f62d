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
//                                              per GPU, equal slabs of a free mode, no exchange
//                [--time ITERS]                print device ms per contraction
//   ctc_contract --permute abcd=dcba --extents a=..,b=.. --dtype f32 --a IN.bin --out OUT.bin
//                                              mode permutation of one tensor (input modes = output modes)
//
// Because the engine runs the command through a shell, write the spec with '='
// instead of '->' inside plugin templates.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
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

int main(int argc, char **argv)
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
    if (ngpu > 1) {
        // Free-mode sharding inside one process (the engine runs ONE command per plugin step): rank r
        // computes slab r of the chosen mode on device r (devices are shared round-robin when the node
        // has fewer than N), and the slabs of C, which are disjoint, are merged on the host.
        if (opt.count("shard")) die("--ngpu and --shard exclude each other");
        const std::string a_modes = spec.substr(0, comma), b_modes = spec.substr(comma + 1, arrow - comma - 1);
        const std::string c_modes = spec.substr(arrow + alen);
        auto is_free = [&](char ch) { return (a_modes.find(ch) != std::string::npos) != (b_modes.find(ch) != std::string::npos); };
        char mode = 0;
        if (opt.count("shard-on") && opt["shard-on"] != "auto") {
            mode = opt["shard-on"][0];
            if (c_modes.find(mode) == std::string::npos || !is_free(mode)) die("--shard-on: not a free mode of this contraction");
        } else {
            for (char ch : c_modes)  // the slowest free mode of C that divides evenly
                if (is_free(ch) && extents[names.find(ch)] % ngpu == 0) { mode = ch; break; }
            if (!mode) die("--ngpu: no free mode whose extent is a multiple of the GPU count");
        }
        const int64_t e_m = extents[names.find(mode)];
        if (e_m % ngpu != 0) die("--ngpu: the extent of the sharded mode must be a multiple of the GPU count");
        size_t s_m = 1;  // stride of the mode in packed C
        for (size_t i = c_modes.find(mode) + 1; i < c_modes.size(); ++i) s_m *= (size_t)extents[names.find(c_modes[i])];
        std::vector<char> hA, hB;
        if (!opt.count("gen-a")) { if (!opt.count("a")) die("--a or --gen-a is required"); hA = read_file(opt["a"], na * es); }
        if (!opt.count("gen-b")) { if (!opt.count("b")) die("--b or --gen-b is required"); hB = read_file(opt["b"], nb * es); }
        std::vector<char> hC(nc * es, 0);
        if (beta != 0.0) hC = read_file(opt["out"], nc * es);
        const int ndev = ctc_tc_device_count();
        std::vector<std::string> errors((size_t)ngpu);
        std::vector<double> rank_ms((size_t)ngpu, 0.0);
        auto worker = [&](int r) {
            auto fail = [&](const std::string &m) { errors[(size_t)r] = m; };
            if (hipSetDevice(r % ndev) != hipSuccess) return fail("hipSetDevice");
            ctc_tc_plan *slab = nullptr;
            int64_t oa = 0, ob = 0, oc = 0;
            if (ctc_tc_plan_shard(plan, mode, r, ngpu, &slab, &oa, &ob, &oc) < 0) return fail(std::string("shard: ") + ctc_tc_last_error());
            void *dA = nullptr, *dB = nullptr, *dC = nullptr;
            if (hipMalloc(&dA, na * es) != hipSuccess || hipMalloc(&dB, nb * es) != hipSuccess || hipMalloc(&dC, nc * es) != hipSuccess)
                return fail("hipMalloc");
            bool ok = true;
            if (hA.empty()) ok = ok && ctc_tc_fill(dA, (int64_t)na, strtoull(opt["gen-a"].c_str(), nullptr, 0), 0, dtype, nullptr) >= 0;
            else ok = ok && hipMemcpy(dA, hA.data(), na * es, hipMemcpyHostToDevice) == hipSuccess;
            if (hB.empty()) ok = ok && ctc_tc_fill(dB, (int64_t)nb, strtoull(opt["gen-b"].c_str(), nullptr, 0), 0, dtype, nullptr) >= 0;
            else ok = ok && hipMemcpy(dB, hB.data(), nb * es, hipMemcpyHostToDevice) == hipSuccess;
            if (beta != 0.0) ok = ok && hipMemcpy(dC, hC.data(), nc * es, hipMemcpyHostToDevice) == hipSuccess;
            if (!ok) return fail("operand upload");
            const void *pa = static_cast<char *>(dA) + oa * es;
            const void *pb = static_cast<char *>(dB) + ob * es;
            void *pc = static_cast<char *>(dC) + oc * es;
            if (opt.count("time")) {
                double ms = 0;
                if (ctc_tc_time_execute(slab, alpha, pa, pb, beta, pc, nullptr, 2, atoi(opt["time"].c_str()), &ms) < 0)
                    return fail(std::string("time: ") + ctc_tc_last_error());
                rank_ms[(size_t)r] = ms;
                if (beta != 0.0 && hipMemcpy(dC, hC.data(), nc * es, hipMemcpyHostToDevice) != hipSuccess) return fail("C upload");
            }
            if (ctc_tc_execute(slab, alpha, pa, pb, beta, pc, nullptr) < 0) return fail(std::string("execute: ") + ctc_tc_last_error());
            if (hipDeviceSynchronize() != hipSuccess) return fail("synchronize");
            // the slab holds the runs [q * e_m + lo, q * e_m + hi) * s_m of C, q = 0, 1, ...: fetch the span
            // that covers them and copy the runs into the host image (ranks touch disjoint elements)
            const size_t lo = (size_t)r * (size_t)(e_m / ngpu), hi = lo + (size_t)(e_m / ngpu);
            const size_t first = lo * s_m, end = nc - ((size_t)e_m - hi) * s_m;
            std::vector<char> span((end - first) * es);
            if (hipMemcpy(span.data(), static_cast<char *>(dC) + first * es, span.size(), hipMemcpyDeviceToHost) != hipSuccess)
                return fail("copy C back");
            const size_t period = (size_t)e_m * s_m, run = (hi - lo) * s_m;
            for (size_t q = 0; q * period < nc; ++q)
                memcpy(hC.data() + (q * period + first) * es, span.data() + q * period * es, run * es);
            ctc_tc_plan_destroy(slab);
            (void)hipFree(dA);
            (void)hipFree(dB);
            (void)hipFree(dC);
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < ngpu; ++r) threads.emplace_back(worker, r);
        for (auto &th : threads) th.join();
        for (int r = 0; r < ngpu; ++r)
            if (!errors[(size_t)r].empty()) die("rank " + std::to_string(r) + ": " + errors[(size_t)r]);
        if (opt.count("time")) {
            double worst = 0;
            for (double ms : rank_ms) worst = ms > worst ? ms : worst;
            int64_t flops = 0;
            ctc_tc_plan_get(plan, CTC_TC_Q_FLOPS, &flops, 1);
            printf("ngpu %d mode %c devices %d max_device_ms %.4f tflops %.3f\n", ngpu, mode, ndev < ngpu ? ndev : ngpu, worst,
                   (double)flops / worst * 1e-9);
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
