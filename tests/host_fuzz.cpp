// host_fuzz.cpp -- sanitizer driver for the host-side C of libsnn_hip.so (test infrastructure, not product).
// Built by `make -C supernnova_amd/csrc asan` together with snn_plan.hip / snn_featurize.hip / snn_h5.hip as plain C++
// under -fsanitize=address,undefined; run by tests/test_host_sanitizers.py.  Every buffer handed to the library is a
// heap block of EXACTLY the size the header promises, so an out-of-bounds access of one byte trips the sanitizer, and
// every hostile input must come back as SNN_OK or a negative status -- never a crash, never an exception.
//
//   host_fuzz_asan plan                      snn_plan_build: edge sizes, hostile lengths
//   host_fuzz_asan featurize                 snn_featurize_plan / _fill: random tables, ties, hostile indices
//   host_fuzz_asan h5 <database.h5> <desc.bin> <rounds>
//        snn_h5_vlen_gather on the real file + descriptor records, then on truncated / bit-flipped copies of both and
//        on descriptors whose heap address is 0, all-ones, or within 16 bytes of 2^64 / of the end of the file
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <random>
#include <string>
#include <vector>

#include "../include/snn_hip.h"

#define REQUIRE(cond)                                                                   \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            fprintf(stderr, "host_fuzz: %s:%d: REQUIRE(%s) failed: %s\n", __FILE__, __LINE__, #cond, snn_last_error()); \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

// exact-size heap block (new[] of n elements: the sanitizer's red zones start right behind it)
template <class T> struct Buf {
    std::unique_ptr<T[]> p;
    size_t n;
    explicit Buf(size_t n_) : p(n_ ? new T[n_]() : nullptr), n(n_) {}
    T* get() { return p.get(); }
};

static int fuzz_plan()
{
    std::mt19937_64 g(1);
    const snn_cfg cfg = {24, 32, 2, 1, 2, SNN_POOL_MEAN, SNN_PREC_F32, SNN_CELL_LSTM};
    for (int n : {1, 2, 31, 32, 33, 63, 64, 65, 1000, 4097}) {
        for (int rep = 0; rep < 4; ++rep) {
            const int n_tiles = (n + 31) / 32;
            Buf<int32_t> len(n), order((size_t)n_tiles * 32), ls((size_t)n_tiles * 32), tT(n_tiles);
            Buf<int64_t> off((size_t)n_tiles + 1);
            const int maxT = rep == 0 ? 1 : rep == 1 ? 300 : 40;
            for (int i = 0; i < n; ++i) len.get()[i] = 1 + (int32_t)(g() % maxT);
            REQUIRE(snn_plan_build(&cfg, len.get(), n, order.get(), ls.get(), tT.get(), off.get()) == SNN_OK);
            // sortedness, permutation, tile lengths, offsets
            std::vector<char> seen(n, 0);
            int64_t want_off = 0;
            for (int t = 0; t < n_tiles; ++t) {
                REQUIRE(off.get()[t] == want_off);
                REQUIRE(tT.get()[t] == ls.get()[t * 32]);
                want_off += (int64_t)tT.get()[t] * 24 * 32;
            }
            REQUIRE(off.get()[n_tiles] == want_off);
            for (int i = 0; i < n_tiles * 32; ++i) {
                const int o = order.get()[i];
                if (i < n) {
                    REQUIRE(o >= 0 && o < n && !seen[o] && ls.get()[i] == len.get()[o]);
                    seen[o] = 1;
                    if (i) REQUIRE(ls.get()[i - 1] >= ls.get()[i]);
                } else REQUIRE(o == -1 && ls.get()[i] == 0);
            }
        }
    }
    // hostile: zero / negative / absurd lengths, n < 1, null pointers, invalid cfg
    Buf<int32_t> len(40), order(64), ls(64), tT(2);
    Buf<int64_t> off(3);
    for (int32_t bad : {0, -1, INT32_MIN, INT32_MAX, SNN_MAX_STEPS + 1}) {
        for (int i = 0; i < 40; ++i) len.get()[i] = 5;
        len.get()[17] = bad;
        REQUIRE(snn_plan_build(&cfg, len.get(), 40, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
    }
    for (int i = 0; i < 40; ++i) len.get()[i] = 5;
    REQUIRE(snn_plan_build(&cfg, len.get(), 0, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
    REQUIRE(snn_plan_build(&cfg, len.get(), -3, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
    REQUIRE(snn_plan_build(&cfg, nullptr, 40, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
    REQUIRE(snn_plan_build(nullptr, len.get(), 40, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
    snn_cfg badcfg = cfg;
    for (int h : {0, -8, 7, 12}) {
        badcfg.hidden_size = h;
        REQUIRE(snn_plan_build(&badcfg, len.get(), 40, order.get(), ls.get(), tT.get(), off.get()) == SNN_ERR_ARG);
        REQUIRE(snn_param_count(&badcfg) < 0 && snn_kparam_count(&badcfg) < 0 && snn_x_step_floats(&badcfg) < 0);
    }
    badcfg = cfg;
    badcfg.num_layers = 99;
    REQUIRE(snn_param_count(&badcfg) < 0);
    printf("plan ok\n");
    return 0;
}

static int fuzz_featurize()
{
    std::mt19937_64 g(2);
    for (int rep = 0; rep < 60; ++rep) {
        const int64_t n_curves = rep < 3 ? rep : 1 + (int64_t)(g() % 50);
        const int64_t n = n_curves == 0 ? 0 : (int64_t)(g() % 3000);
        const int F = 1 + (int)(g() % 6);
        Buf<int64_t> curve(n), order(n), row_of(n), rpc(n_curves), filt(n);
        Buf<double> mjd(n), row_mjd_of(n), flux(n), err(n);
        for (int64_t i = 0; i < n; ++i) {
            curve.get()[i] = (int64_t)(g() % n_curves);
            // coarse times: many exact ties and many gaps around the 8-hour rule
            mjd.get()[i] = 57000.0 + (double)(g() % 200) * (rep % 2 ? 0.11 : 0.5);
            flux.get()[i] = (double)(g() % 1000) - 300.0;
            err.get()[i] = (double)(g() % 7);           // many equal errors: the earliest-on-ties rule is exercised
            filt.get()[i] = (int64_t)(g() % (F + 1)) - 1;   // -1 = "fills nothing"
        }
        const int64_t n_rows = snn_featurize_plan(n, n_curves, curve.get(), mjd.get(), 0.33, order.get(), row_of.get(),
                                                  row_mjd_of.get(), rpc.get());
        REQUIRE(n_rows >= 0 && n_rows <= n);
        int64_t tot = 0;
        for (int64_t c = 0; c < n_curves; ++c) tot += rpc.get()[c];
        REQUIRE(n == 0 || tot == n_rows);
        Buf<double> fo((size_t)n_rows * F), eo((size_t)n_rows * F), rm(n_rows);
        Buf<uint8_t> po((size_t)n_rows * F);
        REQUIRE(snn_featurize_fill(n, n_rows, F, order.get(), row_of.get(), row_mjd_of.get(), flux.get(), err.get(), filt.get(),
                                   fo.get(), eo.get(), po.get(), rm.get()) == SNN_OK);
        if (n > 2) {
            // hostile: a curve index / row / filter / order entry out of range must be refused, not followed
            const int64_t keep = curve.get()[1];
            curve.get()[1] = n_curves;
            REQUIRE(snn_featurize_plan(n, n_curves, curve.get(), mjd.get(), 0.33, order.get(), row_of.get(), row_mjd_of.get(), rpc.get()) == SNN_ERR_ARG);
            curve.get()[1] = -1;
            REQUIRE(snn_featurize_plan(n, n_curves, curve.get(), mjd.get(), 0.33, order.get(), row_of.get(), row_mjd_of.get(), rpc.get()) == SNN_ERR_ARG);
            curve.get()[1] = keep;
            REQUIRE(snn_featurize_plan(n, n_curves, curve.get(), mjd.get(), 0.33, order.get(), row_of.get(), row_mjd_of.get(), rpc.get()) == n_rows);
            const int64_t r1 = row_of.get()[1], o1 = order.get()[1], f1 = filt.get()[order.get()[2]];
            row_of.get()[1] = n_rows;
            REQUIRE(snn_featurize_fill(n, n_rows, F, order.get(), row_of.get(), row_mjd_of.get(), flux.get(), err.get(), filt.get(), fo.get(), eo.get(), po.get(), rm.get()) == SNN_ERR_ARG);
            row_of.get()[1] = r1;
            order.get()[1] = n + 5;
            REQUIRE(snn_featurize_fill(n, n_rows, F, order.get(), row_of.get(), row_mjd_of.get(), flux.get(), err.get(), filt.get(), fo.get(), eo.get(), po.get(), rm.get()) == SNN_ERR_ARG);
            order.get()[1] = -2;
            REQUIRE(snn_featurize_fill(n, n_rows, F, order.get(), row_of.get(), row_mjd_of.get(), flux.get(), err.get(), filt.get(), fo.get(), eo.get(), po.get(), rm.get()) == SNN_ERR_ARG);
            order.get()[1] = o1;
            filt.get()[order.get()[2]] = F;
            REQUIRE(snn_featurize_fill(n, n_rows, F, order.get(), row_of.get(), row_mjd_of.get(), flux.get(), err.get(), filt.get(), fo.get(), eo.get(), po.get(), rm.get()) == SNN_ERR_ARG);
            filt.get()[order.get()[2]] = f1;
        }
    }
    REQUIRE(snn_featurize_plan(-1, 3, nullptr, nullptr, 0.33, nullptr, nullptr, nullptr, nullptr) == SNN_ERR_ARG);
    REQUIRE(snn_featurize_plan(5, 3, nullptr, nullptr, 0.33, nullptr, nullptr, nullptr, nullptr) == SNN_ERR_ARG);
    printf("featurize ok\n");
    return 0;
}

static std::vector<uint8_t> slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "host_fuzz: cannot open %s\n", path); exit(2); }
    std::vector<uint8_t> v;
    uint8_t chunk[65536];
    size_t k;
    while ((k = fread(chunk, 1, sizeof(chunk), f)) > 0) v.insert(v.end(), chunk, chunk + k);
    fclose(f);
    return v;
}

// one call on exact-size copies of (file[:fsize], desc); returns the status
static int gather_once(const std::vector<uint8_t>& file, size_t fsize, const std::vector<uint8_t>& desc, size_t out_bytes)
{
    Buf<uint8_t> fb(fsize), db(desc.size()), ob(out_bytes);
    if (fsize) memcpy(fb.get(), file.data(), fsize);
    if (!desc.empty()) memcpy(db.get(), desc.data(), desc.size());
    return snn_h5_vlen_gather(fb.get(), (int64_t)fsize, db.get(), (int64_t)(desc.size() / 16), 4, ob.get(), (int64_t)out_bytes);
}

static int fuzz_h5(const char* file_path, const char* desc_path, int rounds)
{
    const std::vector<uint8_t> file = slurp(file_path), desc = slurp(desc_path);
    REQUIRE(desc.size() % 16 == 0 && !desc.empty());
    size_t total = 0;
    for (size_t i = 0; i < desc.size(); i += 16) { uint32_t l; memcpy(&l, &desc[i], 4); total += 4 * (size_t)l; }
    REQUIRE(gather_once(file, file.size(), desc, total) == SNN_OK);
    REQUIRE(gather_once(file, file.size(), desc, total - 4) == SNN_ERR_ARG);       // output one element short
    REQUIRE(gather_once(file, 8, desc, total) == SNN_ERR_ARG);                      // file shorter than any header
    std::mt19937_64 g(3);
    int ok = 0, refused = 0;
    auto tally = [&](int rc) { REQUIRE(rc == SNN_OK || rc == SNN_ERR_ARG); (rc == SNN_OK ? ok : refused)++; };
    // descriptors with hostile heap addresses: 0 and all-ones mean "no object" (element skipped), the rest must be refused
    const uint64_t fs = file.size();
    for (uint64_t addr : {(uint64_t)0, ~(uint64_t)0, ~(uint64_t)0 - 1, ~(uint64_t)0 - 15, ~(uint64_t)0 - 16, (uint64_t)1 << 63,
                          fs, fs - 1, fs - 15, fs - 16, fs - 17, (uint64_t)1, (uint64_t)8}) {
        for (size_t k : {(size_t)0, desc.size() / 32 * 16, desc.size() - 16}) {
            std::vector<uint8_t> d = desc;
            memcpy(&d[k + 4], &addr, 8);
            tally(gather_once(file, file.size(), d, total));
        }
    }
    for (int r = 0; r < rounds; ++r) {
        // (a) truncated file
        tally(gather_once(file, (size_t)(g() % (file.size() + 1)), desc, total));
        // (b) bit flips in the descriptor records (length, address, index)
        {
            std::vector<uint8_t> d = desc;
            for (int k = 0, nk = 1 + (int)(g() % 4); k < nk; ++k) d[g() % d.size()] ^= (uint8_t)(1u << (g() % 8));
            tally(gather_once(file, file.size(), d, total));
        }
        // (c) bit flips / byte smashes anywhere in the file (collection headers, object sizes, indices, payload)
        {
            std::vector<uint8_t> f = file;
            for (int k = 0, nk = 1 + (int)(g() % 8); k < nk; ++k) {
                const size_t at = g() % f.size();
                if (g() & 1) f[at] ^= (uint8_t)(1u << (g() % 8));
                else f[at] = (uint8_t)g();
            }
            tally(gather_once(f, f.size(), desc, total));
        }
        // (d) smashed object-size / collection-size fields: 8 random bytes over an 8-aligned word
        {
            std::vector<uint8_t> f = file;
            const size_t at = (g() % (f.size() / 8)) * 8;
            const uint64_t v = g() & 1 ? ~(uint64_t)0 - (g() % 64) : g();
            memcpy(&f[at], &v, 8);
            tally(gather_once(f, f.size(), desc, total));
        }
    }
    printf("h5 ok: %d calls returned SNN_OK, %d were refused\n", ok, refused);
    return 0;
}

int main(int argc, char** argv)
{
    const std::string what = argc > 1 ? argv[1] : "";
    if (what == "plan") return fuzz_plan();
    if (what == "featurize") return fuzz_featurize();
    if (what == "h5" && argc == 5) return fuzz_h5(argv[2], argv[3], atoi(argv[4]));
    fprintf(stderr, "usage: host_fuzz_asan plan | featurize | h5 <database.h5> <desc.bin> <rounds>\n");
    return 2;
}
