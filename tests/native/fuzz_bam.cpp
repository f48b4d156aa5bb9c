// fuzz_bam.cpp -- the native BAM / BGZF / BAI reader (csrc/cv_bam.cpp + cv_inflate.cpp) under AddressSanitizer + UBSan over the
// files named on the command line: a valid BAM written by tests/bam_writer.py and byte-mutated copies of it (and of its index).
// Every file is opened, its references listed, the whole first contig and a region of it read as SAM text and as records
// (with their CIGAR words) until the reader reports the end or an error -- any outcome but a memory error or a hang.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdarg.h>
#include <string>
#include <vector>
#include "../../include/clairvoyante_amd.h"
static thread_local char g_err[512] = "";
void cv_set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof g_err, fmt, ap); va_end(ap); }
extern "C" const char *cv_last_error(void) { return g_err; }

int main(int argc, char **argv)
{
    long opened = 0, lines = 0, recs = 0;
    for (int a = 1; a < argc; a++) {
        for (int threads = 1; threads <= 3; threads += 2) {
            cv_bam *b = nullptr;
            if (cv_bam_open(argv[a], threads, &b) != 0 || !b) continue;
            opened++;
            const int nref = cv_bam_nref(b);
            std::string first;
            for (int i = 0; i < nref && i < 64; i++) {
                const char *name = nullptr; int64_t len = 0;
                if (cv_bam_ref(b, i, &name, &len) == 0 && name && i == 0) first = name;
            }
            (void)cv_bam_has_index(b);
            for (int pass = 0; pass < 4 && !first.empty(); pass++) {
                const bool region = pass & 1, as_records = pass & 2;
                if (cv_bam_view_begin(b, first.c_str(), region ? 500 : 0, region ? 4000 : 0, 2308, pass == 0) != 0) continue;
                int done = 0, guard = 0;
                if (!as_records) {
                    std::vector<char> buf(1 << 16);
                    while (!done && guard++ < 100000) {
                        const int64_t got = cv_bam_view_read(b, buf.data(), (int64_t)buf.size(), &done);
                        if (got < 0) break;
                        for (int64_t k = 0; k < got; k++) lines += buf[(size_t)k] == '\n';
                        if (got == 0 && !done) break;
                    }
                } else {
                    while (!done && guard++ < 100000) {
                        const uint8_t *base = nullptr; const uint32_t *offs = nullptr;
                        const int64_t n = cv_bam_view_records(b, 1 << 18, &base, &offs, &done);
                        if (n < 0) break;
                        for (int64_t i = 0; i < n; i++) {
                            const uint8_t *ops = nullptr; int64_t nops = 0;
                            if (cv_bam_record_cigar(base + offs[i], &ops, &nops) == 0 && nops > 0) {
                                uint32_t w; memcpy(&w, ops + 4 * (nops - 1), 4); recs += (w & 15) < 16;
                            }
                        }
                        if (n == 0 && !done) break;
                    }
                }
            }
            cv_bam_close(b);
        }
    }
    printf("fuzz_bam: %d files, %ld opened, %ld SAM lines, %ld records: no sanitizer report\n", argc - 1, opened, lines, recs);
    return 0;
}
