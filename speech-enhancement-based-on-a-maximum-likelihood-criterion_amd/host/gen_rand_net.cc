// gen_rand_net.cc -- random initial weights in the trainer's .wts format.
// Same command line and initialisation rule as the reference tool
// (Train_code_ML_GGD/pretraining_weights/Gen_rand_net.cpp:63-103):
//   gen_rand_net numlayers l0 l1 ... out_dir out_wts flag beta [seed]
//   flag=1: U(-r, r), r = beta*sqrt(6)/sqrt(n_in+n_out);  flag=0: r = beta/sqrt(n_in); biases 0.
// The uniform draw mirrors the reference's rand()-based integer grid (1e-6 resolution,
// Gen_rand_net.cpp:16-25); an optional seed argument calls srand() (the reference never seeds).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static float uniform_pm1() {
    const int lo = -1000000, hi = 1000000;
    return (float)((rand() % (hi - lo) + lo) / 1000000.0);
}

int main(int argc, char **argv) {
    if (argc < 2) {
        printf("numlayers layersizes[0] ... layersizes[numlayers-1] out_dir out_wts flag beta [seed]\n"
               "flag=0: U(-1/sqrt(n), 1/sqrt(n)) * beta; flag=1: U(+-sqrt(6)/sqrt(n_i+n_j)) * beta\n");
        return 0;
    }
    const int L = atoi(argv[1]);
    if (L < 2 || L > 10 || argc < L + 6) {
        fprintf(stderr, "bad arguments\n");
        return 1;
    }
    std::vector<int> ls(L);
    printf("numlayers=%d\nlayersizes:", L);
    for (int i = 0; i < L; i++) printf("%d, ", ls[i] = atoi(argv[2 + i]));
    printf("\n");
    const std::string out = argv[L + 3];
    const int flag = atoi(argv[L + 4]);
    const float beta = (float)atof(argv[L + 5]);
    if (argc > L + 6) srand((unsigned)atoi(argv[L + 6]));
    FILE *fp = fopen(out.c_str(), "wb");
    if (!fp) {
        fprintf(stderr, "cannot open %s for write\n", out.c_str());
        return 1;
    }
    auto put = [&](const std::string &name, int mrows, int ncols, const std::vector<float> &data) {
        const int32_t stat[5] = {10, mrows, ncols, 0, (int32_t)name.size() + 1};
        fwrite(stat, sizeof(int32_t), 5, fp);
        fwrite(name.c_str(), 1, name.size() + 1, fp);
        fwrite(data.data(), sizeof(float), data.size(), fp);
    };
    for (int i = 1; i < L; i++) {
        const float range = flag ? beta * sqrtf(6.0f) / sqrtf((float)(ls[i - 1] + ls[i])) : beta * 1.0f / sqrtf((float)ls[i - 1]);
        printf("range=%f\n", range);
        std::vector<float> w((size_t)ls[i - 1] * ls[i]), b(ls[i], 0.0f);
        for (float &v : w) v = range * uniform_pm1();
        put("weights" + std::to_string(i) + std::to_string(i + 1), ls[i], ls[i - 1], w);
        put("bias" + std::to_string(i + 1), 1, ls[i], b);
    }
    fclose(fp);
    printf("Saving over!\n");
    return 0;
}
