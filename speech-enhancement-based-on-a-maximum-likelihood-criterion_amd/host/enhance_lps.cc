// enhance_lps.cc -- batch inference: the forward pass of the reference's
// MATLAB decoder (Test_code/decode.m:10-63) on the MI355X engine, without MATLAB:
//   noisy LPS (HTK big-endian, Test_code/readHTK_new.m) -> z-normalise with the training norm
//   file (decode.m:31-33) -> edge-replicated context of fea_context frames
//   (Test_code/frame_expand.m:5-27) -> sigmoid MLP from the trainer's .wts (decode.m:11-18,
//   39-57) -> de-normalise (decode.m:59-61) -> HTK file with sampPeriod 160000, sampSize 4*D,
//   paramKind 9 (decode.m:62, Test_code/writeHTK_new.m:36-51).
// The edge-replicated windows become contiguous slices of a stream padded with (ctx-1)/2 copies
// of the first / last frame, so the forward runs through mlggd_forward_frames.
//
//   enhance_lps wts=mlp.50.wts norm_file=train_noisy.norm in=noisy.lps out=enhanced.htk
//               [fea_context=7] [gpu_used=0] [bunchsize=512] [scp=list of "in out" lines]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/mlggd.h"

namespace {

[[noreturn]] void die(const std::string &m) {
    fprintf(stderr, "enhance_lps: %s\n", m.c_str());
    exit(1);
}
uint32_t bswap32(uint32_t v) { return __builtin_bswap32(v); }
uint16_t bswap16(uint16_t v) { return (uint16_t)((v >> 8) | (v << 8)); }

struct Htk {
    int nframes = 0, samp_period = 0, samp_size = 0, parm_kind = 0;
    std::vector<float> data;  // [nframes][samp_size/4]
};

Htk read_htk(const std::string &path) {  // readHTK_new.m, 'be'
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) die("cannot open " + path);
    uint32_t h[2];
    uint16_t s[2];
    if (fread(h, 4, 2, fp) != 2 || fread(s, 2, 2, fp) != 2) die("short HTK header in " + path);
    Htk f;
    f.nframes = (int)bswap32(h[0]);
    f.samp_period = (int)bswap32(h[1]);
    f.samp_size = bswap16(s[0]);
    f.parm_kind = bswap16(s[1]);
    const size_t n = (size_t)f.nframes * (f.samp_size / 4);
    std::vector<uint32_t> raw(n);
    if (f.nframes <= 0 || f.samp_size % 4 || fread(raw.data(), 4, n, fp) != n) die("bad HTK body in " + path);
    fclose(fp);
    f.data.resize(n);
    for (size_t i = 0; i < n; i++) {
        const uint32_t v = bswap32(raw[i]);
        memcpy(&f.data[i], &v, 4);
    }
    return f;
}

void write_htk(const std::string &path, const float *data, int nframes, int dim) {  // writeHTK_new.m
    FILE *fp = fopen(path.c_str(), "wb");
    if (!fp) die("cannot open " + path + " for writing");
    const uint32_t h[2] = {bswap32((uint32_t)nframes), bswap32(160000u)};
    const uint16_t s[2] = {bswap16((uint16_t)(dim * 4)), bswap16(9)};
    fwrite(h, 4, 2, fp);
    fwrite(s, 2, 2, fp);
    std::vector<uint32_t> raw((size_t)nframes * dim);
    for (size_t i = 0; i < raw.size(); i++) {
        uint32_t v;
        memcpy(&v, &data[i], 4);
        raw[i] = bswap32(v);
    }
    fwrite(raw.data(), 4, raw.size(), fp);
    fclose(fp);
}

}  // namespace

int main(int argc, char **argv) {
    std::string wts, norm_file, in, out, scp;
    int ctx = 7, gpu = 0, bunch = 512;
    for (int a = 1; a < argc; a++) {
        const std::string arg(argv[a]);
        const size_t eq = arg.find('=');
        if (eq == std::string::npos) die("Arg: " + arg + "  Format Error");
        const std::string k = arg.substr(0, eq), v = arg.substr(eq + 1);
        if (k == "wts") wts = v;
        else if (k == "norm_file") norm_file = v;
        else if (k == "in") in = v;
        else if (k == "out") out = v;
        else if (k == "scp") scp = v;
        else if (k == "fea_context") ctx = atoi(v.c_str());
        else if (k == "gpu_used") gpu = atoi(v.c_str());
        else if (k == "bunchsize") bunch = atoi(v.c_str());
    }
    if (wts.empty() || norm_file.empty() || (scp.empty() && (in.empty() || out.empty())))
        die("usage: enhance_lps wts=F norm_file=F (in=F out=F | scp=LIST) [fea_context=7] [gpu_used=0] [bunchsize=512]");
    if (ctx < 1 || ctx % 2 == 0) die("fea_context must be odd");

    // ---- model: the trainer's .wts container (Interface.cc:484-516)
    std::vector<std::vector<float>> W(1), Bv(1);
    std::vector<int> ls;
    {
        FILE *fp = fopen(wts.c_str(), "rb");
        if (!fp) die("cannot open " + wts);
        int32_t stat[5];
        char name[256];
        while (fread(stat, 4, 5, fp) == 5) {
            if (stat[4] < 1 || stat[4] > 255 || fread(name, 1, stat[4], fp) != (size_t)stat[4]) die("bad matrix header in " + wts);
            std::vector<float> m((size_t)stat[1] * stat[2]);
            if (fread(m.data(), 4, m.size(), fp) != m.size()) die("truncated matrix in " + wts);
            if (stat[1] != 1) {  // weights: mrows = out, ncols = in
                if (ls.empty()) ls.push_back(stat[2]);
                if (ls.back() != stat[2]) die("layer sizes in " + wts + " do not chain");
                ls.push_back(stat[1]);
                W.push_back(m);
            } else {
                Bv.push_back(m);
            }
        }
        fclose(fp);
        if (W.size() < 2 || W.size() != Bv.size() || (int)W.size() > MLGGD_MAXLAYER) die("unexpected matrix list in " + wts);
    }
    const int L = (int)ls.size(), D = ls[L - 1];
    if (ls[0] % ctx) die("layersizes[0] is not a multiple of fea_context");
    const int dim = ls[0] / ctx;

    // ---- norm file (Interface.cc:373-399 layout: "vec N", N means, "vec N", N inverse std-devs)
    std::vector<float> mean(dim), inv(dim);
    {
        std::ifstream f(norm_file);
        if (!f) die("cannot open " + norm_file);
        std::string line;
        std::getline(f, line);
        for (int j = 0; j < dim; j++) { std::getline(f, line); mean[j] = (float)atof(line.c_str()); }
        std::getline(f, line);
        for (int j = 0; j < dim; j++) { std::getline(f, line); inv[j] = (float)atof(line.c_str()); }
    }
    if (D % dim) die("output dimension is not a multiple of the feature dimension");

    mlggd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.device = gpu;
    cfg.numlayers = L;
    for (int i = 0; i < L; i++) cfg.layersizes[i] = ls[i];
    cfg.bunchsize = bunch;
    cfg.shapefactor = 2.0f;
    std::vector<const float *> wp(L, nullptr), bp(L, nullptr);
    for (int l = 1; l < L; l++) { wp[l] = W[l].data(); bp[l] = Bv[l].data(); }
    mlggd_handle h = nullptr;
    if (mlggd_create(&cfg, wp.data(), bp.data(), &h) != MLGGD_OK) die(std::string("mlggd_create: ") + mlggd_last_error());

    std::vector<std::pair<std::string, std::string>> jobs;
    if (!scp.empty()) {
        std::ifstream f(scp);
        if (!f) die("cannot open " + scp);
        std::string a, b;
        while (f >> a >> b) jobs.emplace_back(a, b);
    } else {
        jobs.emplace_back(in, out);
    }
    const int half = (ctx - 1) / 2;
    for (const auto &job : jobs) {
        const Htk x = read_htk(job.first);
        if (x.samp_size != dim * 4) die(job.first + ": feature dimension does not match the model");
        const int n = x.nframes, np = n + 2 * half;
        std::vector<float> stream((size_t)np * dim);
        for (int t = 0; t < np; t++) {  // frame_expand.m: clamp to the first / last frame
            int src = t - half;
            src = src < 0 ? 0 : (src >= n ? n - 1 : src);
            for (int j = 0; j < dim; j++) stream[(size_t)t * dim + j] = (x.data[(size_t)src * dim + j] - mean[j]) * inv[j];
        }
        std::vector<int32_t> first(n);
        for (int t = 0; t < n; t++) first[t] = t;
        std::vector<float> y((size_t)n * D);
        if (mlggd_forward_frames(h, np, ctx, stream.data(), n, first.data(), y.data()) != MLGGD_OK)
            die(std::string("mlggd_forward_frames: ") + mlggd_last_error());
        for (int t = 0; t < n; t++)
            for (int j = 0; j < D; j++) y[(size_t)t * D + j] = y[(size_t)t * D + j] / inv[j % dim] + mean[j % dim];  // decode.m:59-61
        write_htk(job.second, y.data(), n, D);
        printf("%s -> %s (%d frames)\n", job.first.c_str(), job.second.c_str(), n);
    }
    mlggd_destroy(h);
    return 0;
}
