// fuzz_host.cpp -- the host data plane (csrc/cv_hostio.cpp, cv_inflate.cpp: text-tensor parser, blosc/LZ4 decoder, inflate,
// VCF formatter) under AddressSanitizer + UndefinedBehaviorSanitizer on the CPU (GPU sanitizers are not available on
// this pool).  Valid inputs are built first -- tensor rows in CreateTensor.py's text format (dataPrepScripts/
// CreateTensor.py:52-56), blosc chunks from the library's own encoder, deflate streams from zlib -- decoded and checked,
// then byte-mutated / truncated copies are fed to the same entry points: any result is allowed except a memory error, a
// hang or undefined behaviour.  Every buffer is allocated at its exact size so that an over-read is caught.
//   make -C tests/native        (g++ -fsanitize=address,undefined; run by tests/test_native_sanitizers.py)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>
#include <string>
#include <vector>
#include "../../include/clairvoyante_amd.h"

// (csrc/cv_api.hip holds the library's error slot; the harness links the host sources alone)
#include <stdarg.h>
static thread_local char g_err[512] = "";
void cv_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char *cv_last_error(void) { return g_err; }

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static int rint_(int n) { return (int)(rnd() % (uint64_t)n); }
#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "fuzz_host: %s:%d: %s failed\n", __FILE__, __LINE__, #c); exit(1); } } while (0)

static std::vector<uint8_t> exact(const std::string &s) { return std::vector<uint8_t>(s.begin(), s.end()); }

static void mutate(std::vector<uint8_t> &b)
{
    if (b.empty()) return;
    switch (rint_(6)) {
    case 0: for (int k = rint_(4) + 1; k > 0; k--) b[rint_((int)b.size())] = (uint8_t)rnd(); break;       // random bytes
    case 1: b.resize(rint_((int)b.size()) + 1); break;                                                    // truncate
    case 2: for (int k = rint_(3) + 1; k > 0; k--) b[rint_((int)b.size())] ^= (uint8_t)(1u << rint_(8)); break;   // bit flips
    case 3: { int at = rint_((int)b.size()), len = rint_(64) + 1; for (int i = at; i < at + len && i < (int)b.size(); i++) b[i] = 0xff; } break;
    case 4: { int at = rint_((int)b.size()), len = rint_(64) + 1; for (int i = at; i < at + len && i < (int)b.size(); i++) b[i] = 0; } break;
    default: { int at = rint_((int)b.size()); int len = rint_(32) + 1; std::vector<uint8_t> ins(len); for (auto &v : ins) v = (uint8_t)rnd(); b.insert(b.begin() + at, ins.begin(), ins.end()); } break;
    }
}

// ---- text tensors -------------------------------------------------------------------------------------------------
static std::string tensor_rows(int rows)
{
    std::string s; char num[32];
    static const char bases[] = "ACGTN";
    for (int r = 0; r < rows; r++) {
        s += "chr21 "; snprintf(num, sizeof num, "%d ", 1000 + rint_(1 << 28)); s += num;
        for (int i = 0; i < 33; i++) s += bases[rint_(rint_(50) ? 4 : 5)];
        for (int i = 0; i < 528; i++) { snprintf(num, sizeof num, " %d.%d", rint_(3) ? 0 : rint_(60), rint_(10)); s += num; }
        s += "\n";
    }
    return s;
}

static void parse_text(const std::vector<uint8_t> &buf, int64_t max_rows, bool must_parse, int64_t want_rows)
{
    std::vector<float> x((size_t)max_rows * 528 + 1);
    std::vector<int64_t> meta((size_t)max_rows * 6 + 1);
    int64_t consumed = -1, nrows = -1, nbad = -1;
    // the parser takes (pointer, length): no terminating byte after the buffer
    const int rc = cv_parse_tensor_text((const char *)buf.data(), (int64_t)buf.size(), max_rows, x.data(), meta.data(), &consumed, &nrows, &nbad);
    if (must_parse) { REQUIRE(rc == 0); REQUIRE(nrows + nbad >= 0 && nrows <= want_rows); REQUIRE(consumed <= (int64_t)buf.size()); }
    if (rc == 0) {
        REQUIRE(nrows >= 0 && nrows <= max_rows && consumed >= 0 && consumed <= (int64_t)buf.size());
        for (int64_t r = 0; r < nrows; r++)
            for (int k = 0; k < 6; k += 2) REQUIRE(meta[r * 6 + k] >= 0 && meta[r * 6 + k] + meta[r * 6 + k + 1] <= (int64_t)buf.size());
    }
}

// ---- VCF records ----------------------------------------------------------------------------------------------------
// rows parsed from text (x, meta), decisions as cv_call_postproc writes them -- in range first, then with out-of-range words
static void vcf_round(int rows, bool wild)
{
    std::vector<uint8_t> text = exact(tensor_rows(rows));
    std::vector<float> x((size_t)rows * 528);
    std::vector<int64_t> meta((size_t)rows * 6);
    int64_t consumed = 0, nrows = 0, nbad = 0;
    REQUIRE(cv_parse_tensor_text((const char *)text.data(), (int64_t)text.size(), rows, x.data(), meta.data(), &consumed, &nrows, &nbad) == 0);
    if (nrows == 0) return;
    const int64_t n = nrows;
    x.resize((size_t)n * 528); x.shrink_to_fit(); meta.resize((size_t)n * 6); meta.shrink_to_fit();
    std::vector<int32_t> call((size_t)n * 8, 0);
    std::vector<float> qual((size_t)n * 4, 0.f);
    std::vector<int64_t> xrow((size_t)n), prow((size_t)n);
    for (int64_t i = 0; i < n; i++) {
        int32_t *c = &call[(size_t)i * 8];
        c[0] = rint_(4); c[1] = rint_(2); c[2] = rint_(6); c[3] = rint_(4); c[4] = rint_(4);
        qual[(size_t)i * 4 + 0] = (float)rint_(1000) / 1000.f; qual[(size_t)i * 4 + 1] = (float)rint_(1000) / 1000.f * qual[(size_t)i * 4];
        qual[(size_t)i * 4 + 2] = rint_(8) ? (float)rint_(200) : 0.f;
        xrow[(size_t)i] = rint_((int)n); prow[(size_t)i] = rint_((int)n);
        if (wild) {
            if (!rint_(3)) c[rint_(5)] = (int32_t)rnd();
            if (!rint_(4)) { uint32_t bits = (uint32_t)rnd(); memcpy(&qual[(size_t)i * 4 + rint_(3)], &bits, 4); }
        }
    }
    for (int pass = 0; pass < 2; pass++) {
        const bool rowsel = pass == 1;
        int64_t need = 0, nrec = 0;
        char dummy;
        int rc = cv_format_vcf(call.data(), qual.data(), n, x.data(), rowsel ? xrow.data() : nullptr, (const char *)text.data(), meta.data(),
                               rowsel ? prow.data() : nullptr, rint_(2), rint_(2), rint_(100), &dummy, 0, &need, &nrec);
        if (wild && rc == 1) continue;               // out-of-range decisions may be refused
        REQUIRE(rc == 2 || (rc == 0 && need == 0));
        if (rc == 0) continue;
        // (the options were random: ask again with fixed ones for the two sized calls)
        rc = cv_format_vcf(call.data(), qual.data(), n, x.data(), rowsel ? xrow.data() : nullptr, (const char *)text.data(), meta.data(),
                           rowsel ? prow.data() : nullptr, 1, 1, 30, &dummy, 0, &need, &nrec);
        if (wild && rc == 1) continue;
        REQUIRE(rc == 2 || (rc == 0 && need == 0));
        if (rc == 0) continue;
        std::vector<char> out((size_t)need);
        int64_t len = 0;
        rc = cv_format_vcf(call.data(), qual.data(), n, x.data(), rowsel ? xrow.data() : nullptr, (const char *)text.data(), meta.data(),
                           rowsel ? prow.data() : nullptr, 1, 1, 30, out.data(), need, &len, &nrec);
        REQUIRE(rc == 0 && len == need && nrec <= n);
        if (need > 1) {
            std::vector<char> small((size_t)need - 1);
            REQUIRE(cv_format_vcf(call.data(), qual.data(), n, x.data(), rowsel ? xrow.data() : nullptr, (const char *)text.data(), meta.data(),
                                  rowsel ? prow.data() : nullptr, 1, 1, 30, small.data(), need - 1, &len, &nrec) == 2);
        }
    }
}

// ---- blosc ----------------------------------------------------------------------------------------------------------
static void blosc_round(const std::vector<uint8_t> &chunk, const std::vector<uint8_t> *want)
{
    const int64_t nb = cv_blosc_nbytes(chunk.data(), (int64_t)chunk.size());
    if (nb < 0 || nb > (64 << 20)) { REQUIRE(!want); return; }
    std::vector<uint8_t> dst((size_t)nb);
    const int rc = cv_blosc_decompress(chunk.data(), (int64_t)chunk.size(), dst.data(), nb);
    if (want) { REQUIRE(rc == 0 && nb == (int64_t)want->size() && memcmp(dst.data(), want->data(), (size_t)nb) == 0); }
    if (nb > 8) {          // a destination that is too small must be refused, not overrun
        std::vector<uint8_t> small((size_t)nb - 7);
        (void)cv_blosc_decompress(chunk.data(), (int64_t)chunk.size(), small.data(), nb - 7);
    }
    // the block form: the chunk as a pickled ndarray (most mutated chunks are "layout not recognised")
    const uint8_t *cp = chunk.data(); const int64_t cl = (int64_t)chunk.size();
    std::vector<uint8_t> blk((size_t)nb + 16); int64_t len = 0; int32_t st = 0;
    (void)cv_blosc_unpack_blocks(&cp, &cl, 1, blk.data(), nb, &len, &st);
}

// several chunks per call on the host threads; the blocks hold a pickled ndarray as blosc.pack_array writes it (protocol 2
// header, BINBYTES 'B' + u32 length + data, short trailer), one of them mutated
static void many_round()
{
    const int nb = 3 + rint_(10);
    const int64_t block_bytes = 4 * (int64_t)(500 + rint_(2000));
    std::vector<std::vector<uint8_t>> raw((size_t)nb), chunks((size_t)nb);
    const int hurt = rint_(2) ? nb : rint_(nb);            // half of the calls hold no damaged chunk
    std::vector<int64_t> dlen((size_t)nb);
    for (int i = 0; i < nb; i++) {
        const int64_t data = i == nb - 1 ? 4 * (int64_t)(1 + rint_((int)(block_bytes / 4))) : block_bytes;
        dlen[(size_t)i] = data;
        std::vector<uint8_t> &r = raw[(size_t)i];
        static const char head[] = "\x80\x02cnumpy.core.multiarray\n_reconstruct\nq\x00";
        r.assign(head, head + sizeof head - 1);
        r.push_back('B'); for (int k = 0; k < 4; k++) r.push_back((uint8_t)((uint64_t)data >> (8 * k)));
        for (int64_t k = 0; k < data; k++) r.push_back((uint8_t)(k % 7 ? 0 : rnd()));
        static const char tail[] = "q\x01tq\x02b.";
        r.insert(r.end(), tail, tail + sizeof tail - 1);
        std::vector<uint8_t> &c = chunks[(size_t)i];
        c.resize(r.size() + r.size() / 200 + 256); int64_t clen = 0;
        REQUIRE(cv_blosc_compress_lz4(r.data(), (int64_t)r.size(), 4, c.data(), (int64_t)c.size(), &clen) == 0);
        c.resize((size_t)clen); c.shrink_to_fit();
        if (i == hurt) { for (int m = 3; m > 0; m--) mutate(c); c.shrink_to_fit(); }
    }
    std::vector<const uint8_t *> cp((size_t)nb); std::vector<int64_t> cl((size_t)nb), lens((size_t)nb), caps((size_t)nb);
    std::vector<int32_t> st((size_t)nb);
    std::vector<std::vector<uint8_t>> outs((size_t)nb); std::vector<uint8_t *> op((size_t)nb);
    for (int i = 0; i < nb; i++) {
        cp[(size_t)i] = chunks[(size_t)i].data(); cl[(size_t)i] = (int64_t)chunks[(size_t)i].size();
        const int64_t want = cv_blosc_nbytes(cp[(size_t)i], cl[(size_t)i]);
        caps[(size_t)i] = want >= 0 && want < (64 << 20) ? want : 0;
        outs[(size_t)i].resize((size_t)caps[(size_t)i] + 1); op[(size_t)i] = outs[(size_t)i].data();
    }
    const int rc = cv_blosc_decompress_many(cp.data(), cl.data(), op.data(), caps.data(), nb, st.data());
    for (int i = 0; i < nb; i++)
        if (i != hurt) REQUIRE(st[(size_t)i] == 0 && memcmp(op[(size_t)i], raw[(size_t)i].data(), raw[(size_t)i].size()) == 0);
    if (hurt >= nb) REQUIRE(rc == 0);
    std::vector<uint8_t> dst((size_t)(block_bytes * nb));
    const int rc2 = cv_blosc_unpack_blocks(cp.data(), cl.data(), nb, dst.data(), block_bytes, lens.data(), st.data());
    if (hurt >= nb) {
        REQUIRE(rc2 == 0);
        for (int i = 0; i < nb; i++) REQUIRE(lens[(size_t)i] == dlen[(size_t)i]);
    }
}

// ---- inflate --------------------------------------------------------------------------------------------------------
static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &src, int level)
{
    z_stream z; memset(&z, 0, sizeof z);
    REQUIRE(deflateInit2(&z, level, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) == Z_OK);
    std::vector<uint8_t> out(deflateBound(&z, src.size()) + 16);
    z.next_in = (Bytef *)src.data(); z.avail_in = (uInt)src.size(); z.next_out = out.data(); z.avail_out = (uInt)out.size();
    REQUIRE(deflate(&z, Z_FINISH) == Z_STREAM_END);
    out.resize(z.total_out); deflateEnd(&z);
    return out;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    if (argc > 2) rng_state ^= (uint64_t)atoll(argv[2]) * 0x2545F4914F6CDD1Dull;
    cv_set_host_threads(1);
    long n_text = 0, n_blosc = 0, n_inf = 0, n_vcf = 0;
    for (int it = 0; it < iters; it++) {
        if (it % 50 == 0) cv_set_host_threads(1 + rint_(4));
        // text rows
        {
            const int rows = 1 + rint_(6);
            std::vector<uint8_t> good = exact(tensor_rows(rows));
            parse_text(good, rows + 2, true, rows);
            parse_text(good, 1, true, rows);
            for (int k = 0; k < 6; k++) { std::vector<uint8_t> b = good; for (int m = rint_(3) + 1; m > 0; m--) mutate(b); parse_text(b, rows + 2, false, 0); n_text++; }
        }
        if (it % 40 == 7) {        // buffers large enough for the parser's and the formatter's host threads
            cv_set_host_threads(2 + rint_(6));
            const int rows = 2100 + rint_(400);
            std::vector<uint8_t> big = exact(tensor_rows(rows));
            parse_text(big, rows, true, rows);
            parse_text(big, rows / 3, true, rows);
            for (int k = 0; k < 3; k++) { std::vector<uint8_t> b = big; for (int m = 40; m > 0; m--) mutate(b); parse_text(b, rows + 8, false, 0); n_text++; }
            vcf_round(rows, false); vcf_round(rows, true);
            many_round();
        }
        vcf_round(1 + rint_(40), false);
        vcf_round(1 + rint_(40), true); n_vcf += 2;
        // blosc chunks
        {
            const int typesize = 1 << rint_(4);      // 1, 2, 4, 8
            const int n = (1 + rint_(4000)) * typesize;
            std::vector<uint8_t> src((size_t)n);
            const int kind = rint_(3);
            for (int i = 0; i < n; i++) src[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)(i / 97) : (uint8_t)(rint_(8) ? 0 : rnd());
            std::vector<uint8_t> chunk((size_t)n + n / 200 + 256); int64_t clen = 0;
            REQUIRE(cv_blosc_compress_lz4(src.data(), n, typesize, chunk.data(), (int64_t)chunk.size(), &clen) == 0);
            chunk.resize((size_t)clen);
            blosc_round(chunk, &src);
            for (int k = 0; k < 8; k++) { std::vector<uint8_t> b = chunk; for (int m = rint_(3) + 1; m > 0; m--) mutate(b); blosc_round(b, nullptr); n_blosc++; }
        }
        // deflate streams
        {
            const int n = 1 + rint_(20000);
            std::vector<uint8_t> src((size_t)n);
            const int kind = rint_(3);
            for (int i = 0; i < n; i++) src[i] = kind == 0 ? (uint8_t)rnd() : kind == 1 ? (uint8_t)("ACGT \n0123"[rint_(10)]) : (uint8_t)(i % 251);
            // (the decoder's contract, include/clairvoyante_amd.h: src[0, n) is followed by 8 readable bytes -- the gzip /
            // BGZF trailer; exactly 8 are appended here, so a read past them is caught)
            std::vector<uint8_t> comp = deflate_raw(src, rint_(10));
            const int64_t clen = (int64_t)comp.size();
            comp.resize(comp.size() + 8, 0xa5);
            std::vector<uint8_t> dst((size_t)n);
            REQUIRE(cv_inflate_raw(comp.data(), clen, dst.data(), n) == n && memcmp(dst.data(), src.data(), (size_t)n) == 0);
            if (n > 4) { std::vector<uint8_t> small((size_t)n - 3); REQUIRE(cv_inflate_raw(comp.data(), clen, small.data(), n - 3) < 0); }
            {       // the streaming form over the same data, in pieces of at least `want` bytes
                std::vector<uint8_t> d3((size_t)n); int64_t bitpos = 0, have = 0; int32_t fin = 0; int guard = 0;
                while (!fin && guard++ < 100000) {
                    const int64_t got = cv_inflate_stream(comp.data(), clen, &bitpos, d3.data(), have, n, 1 + rint_(4096), &fin);
                    REQUIRE(got >= 0); have += got;
                }
                REQUIRE(fin && have == n && memcmp(d3.data(), src.data(), (size_t)n) == 0);
            }
            for (int k = 0; k < 8; k++) {
                std::vector<uint8_t> b(comp.begin(), comp.begin() + clen); for (int m = rint_(3) + 1; m > 0; m--) mutate(b);
                const int64_t bl = (int64_t)b.size();
                b.resize(b.size() + 8, 0x5a);
                std::vector<uint8_t> d2((size_t)n);
                const int64_t got = cv_inflate_raw(b.data(), bl, d2.data(), n);
                REQUIRE(got <= n);
                int64_t bitpos = 0, have = 0; int32_t fin = 0; int guard = 0;
                while (!fin && guard++ < 100000) {
                    const int64_t g2 = cv_inflate_stream(b.data(), bl, &bitpos, d2.data(), have, n, 1 + rint_(4096), &fin);
                    if (g2 < 0) break;
                    have += g2; REQUIRE(have <= n);
                    if (g2 == 0 && !fin) break;
                }
                n_inf++;
            }
        }
    }
    printf("fuzz_host: %d rounds, %ld mutated text buffers, %ld mutated blosc chunks, %ld mutated deflate streams, %ld batches of VCF records: no sanitizer report\n", iters, n_text, n_blosc, n_inf, n_vcf);
    return 0;
}
