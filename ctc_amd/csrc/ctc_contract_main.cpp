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
                   "       [--shard MODE:RANK/NRANKS] [--time ITERS] [--device N]\n"
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
