// dp_launch.cc -- see dp_launch.h
#include "dp_launch.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <unistd.h>

#include "trainer_io.h"  // IoError

namespace mlggd_host {

std::vector<int> rank_sample_rows(int n_samples, int bunch, int world, int rank) {
    std::vector<int> rows;
    if (n_samples <= 0 || bunch <= 0 || world <= 0 || rank < 0 || rank >= world) return rows;
    const int gb = bunch * world, nglob = n_samples / gb;
    rows.resize((size_t)nglob * bunch);
    for (int g = 0; g < nglob; g++)
        for (int j = 0; j < bunch; j++) rows[(size_t)g * bunch + j] = g * gb + rank * bunch + j;
    return rows;
}

namespace {

const char kMagic[8] = {'M', 'L', 'G', 'G', 'D', 'I', 'D', '2'};

double now_s() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

uint64_t fresh_nonce() {
    uint64_t v = 0;
    if (FILE *fp = fopen("/dev/urandom", "rb")) {
        if (fread(&v, 1, sizeof(v), fp) != sizeof(v)) v = 0;
        fclose(fp);
    }
    if (v == 0) {  // no urandom: clock + pid still differs from any earlier launch
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        v = ((uint64_t)ts.tv_sec << 32) ^ (uint64_t)ts.tv_nsec ^ ((uint64_t)getpid() << 48) ^ 0x9e3779b97f4a7c15ull;
    }
    return v ? v : 1;
}

void write_atomic(const std::string &path, const void *data, size_t bytes) {
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) throw IoError("cannot write " + tmp);
    const bool ok = fwrite(data, 1, bytes, fp) == bytes;
    if (fclose(fp) != 0 || !ok || rename(tmp.c_str(), path.c_str()) != 0) {
        unlink(tmp.c_str());
        throw IoError("cannot publish " + path);
    }
}

bool read_exact(const std::string &path, void *data, size_t bytes) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) return false;
    unsigned char extra;
    const bool ok = fread(data, 1, bytes, fp) == bytes && fread(&extra, 1, 1, fp) == 0;
    fclose(fp);
    return ok;
}

struct IdFile {
    char magic[8];
    uint64_t nonce;
    unsigned char id[kUniqueIdBytes];
};
struct AckFile {
    uint64_t nonce_seen, mine;
};

std::string ack_path(const std::string &path, int r) { return path + ".ack." + std::to_string(r); }

}  // namespace

void rendezvous_cleanup(const std::string &path, int world) {
    unlink(path.c_str());
    unlink((path + ".go").c_str());
    for (int r = 1; r < world; r++) unlink(ack_path(path, r).c_str());
}

void rendezvous(const std::string &path, int world, int rank, unsigned char id[kUniqueIdBytes], double timeout_s) {
    if (path.empty()) throw IoError("WORLD_SIZE > 1 needs MLGGD_ID_FILE (path visible to every rank)");
    if (world < 2 || rank < 0 || rank >= world) throw IoError("rendezvous: bad world/rank");
    const double t_end = now_s() + timeout_s;
    const std::string go = path + ".go";
    if (rank == 0) {
        rendezvous_cleanup(path, world);  // whatever an earlier (crashed) launch left behind
        IdFile f;
        memcpy(f.magic, kMagic, 8);
        f.nonce = fresh_nonce();
        memcpy(f.id, id, kUniqueIdBytes);
        write_atomic(path, &f, sizeof(f));
        std::vector<uint64_t> gof(world, 0);
        gof[0] = f.nonce;
        for (;;) {
            int have = 0;
            for (int r = 1; r < world; r++) {
                AckFile a;
                if (read_exact(ack_path(path, r), &a, sizeof(a)) && a.nonce_seen == f.nonce) {
                    gof[r] = a.mine;
                    have++;
                }
            }
            if (have == world - 1) break;
            if (now_s() > t_end) throw IoError("timed out waiting for the other ranks at MLGGD_ID_FILE");
            usleep(5000);
        }
        write_atomic(go, gof.data(), gof.size() * sizeof(uint64_t));
        return;
    }
    const uint64_t mine = fresh_nonce();
    uint64_t acked = 0;
    IdFile cur;
    memset(&cur, 0, sizeof(cur));
    for (;;) {
        IdFile f;
        if (read_exact(path, &f, sizeof(f)) && memcmp(f.magic, kMagic, 8) == 0 && f.nonce != 0 && f.nonce != acked) {
            const AckFile a = {f.nonce, mine};
            write_atomic(ack_path(path, rank), &a, sizeof(a));
            acked = f.nonce;
            cur = f;
        }
        if (acked != 0) {
            std::vector<uint64_t> gof(world, 0);
            if (read_exact(go, gof.data(), gof.size() * sizeof(uint64_t)) && gof[0] == acked && gof[rank] == mine) {
                memcpy(id, cur.id, kUniqueIdBytes);
                return;
            }
        }
        if (now_s() > t_end) throw IoError("timed out waiting for MLGGD_ID_FILE");
        usleep(5000);
    }
}

}  // namespace mlggd_host
