// cv_bam.cpp -- native `samtools view -F <mask> BAM CTG[:S-E]` (host code, no GPU): BGZF blocks inflated by several
// threads (zlib), BAM records of the region turned into the SAM text lines the pileup parser reads.
//
// The reference shells out to `samtools view` for every stage and every chunk (dataPrepScripts/CreateTensor.py:128-130,
// ExtractVariantCandidates.py:112-114); with the pileup itself on the GPU that single-threaded decoder is what bounds a
// whole-genome run.  This is the same stream without the external process: an OPTIONAL producer (--samtools native);
// the default remains the samtools pipe.  Formats follow the SAM/BAM specification (SAMv1 sections 4.1 BGZF, 4.2 BAM,
// 5.2 BAI): nothing of it exists in the reference tree, so the reader is validated against BAM files written by
// tests/bam_writer.py from the same SAM text -- "parity unpinned" against htslib itself.
//
// Printed per record (what the consumers read): QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL, QUAL as '*'
// unless asked for; auxiliary tags are not printed.  A record with more than 65535 CIGAR operations keeps the
// placeholder <l_seq>S<span>N inline and the real operations in its CG:B,I tag (SAMv1 4.2.2): the real ones are used.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <string>
#include <thread>
#include <vector>

#include "../../include/clairvoyante_amd.h"

void cv_set_error(const char *fmt, ...);

namespace {

inline uint32_t rd_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t rd_i32(const uint8_t *p) { return (int32_t)rd_u32(p); }
inline uint16_t rd_u16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint64_t rd_u64(const uint8_t *p) { return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); }

struct ref_t { std::string name; int64_t len; };

// growable byte buffer without value-initialisation (std::vector<uint8_t>::resize zero-fills every inflated
// byte before the decoder overwrites it, and re-copies everything when it grows: that cost as much as inflating)
struct bytebuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    ~bytebuf() { free(p); }
    bytebuf() = default;
    bytebuf(const bytebuf &) = delete;
    bytebuf &operator=(const bytebuf &) = delete;
    uint8_t *data() { return p; }
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    void clear() { n = 0; }
    bool resize(size_t want)                                   // new bytes are uninitialised; false = out of memory
    {
        if (want > cap) {
            size_t c = cap ? cap : (size_t)1 << 20;
            while (c < want) c += c / 2 + ((size_t)1 << 20);
            uint8_t *q = (uint8_t *)realloc(p, c);
            if (!q) return false;
            p = q; cap = c;
        }
        n = want;
        return true;
    }
    void drop_front(size_t k) { memmove(p, p + k, n - k); n -= k; }
};

}  // namespace

struct cv_bam {
    FILE *fp = nullptr;
    std::string path;
    std::vector<ref_t> refs;
    int64_t first_record_voff = 0;           // virtual offset of the first alignment record
    bool has_index = false;
    std::vector<std::vector<uint64_t>> linear;   // per reference: BAI linear index (16 kb windows)
    // ---- current view
    int tid = -1;
    int64_t beg0 = 0, end0 = 0;              // 0-based half-open region
    int exclude = 0, with_qual = 0, threads = 1;
    bool done = true;
    int64_t next_coff = 0;                   // file offset of the next BGZF block to read
    bytebuf data;                            // inflated bytes not yet consumed
    size_t data_pos = 0;
    bool eof = false;
    bytebuf comp;                            // scratch: compressed blocks of one batch
    std::vector<uint32_t> rec_offs;          // cv_bam_view_records: starts of the selected records in `data`
};

namespace {

// upper bound on a record's block_size (SAMv1 4.2 gives none; the longest reads in practice are a few Mbp)
const int64_t kMaxRecord = (int64_t)1 << 30;

// one BGZF block at p (n bytes available): total size, or 0 if incomplete, or -1 if not a BGZF block
int bgzf_block_size(const uint8_t *p, size_t n)
{
    if (n < 18) return 0;
    if (p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || !(p[3] & 4)) return -1;
    const int xlen = rd_u16(p + 10);
    if (n < (size_t)12 + xlen) return 0;
    int off = 12, bsize = -1;
    while (off + 4 <= 12 + xlen) {
        const int slen = rd_u16(p + off + 2);
        if (p[off] == 'B' && p[off + 1] == 'C' && slen == 2) {
            if (off + 6 > 12 + xlen) return -1;            // BC payload would lie behind the extra field
            bsize = rd_u16(p + off + 4) + 1;
        }
        off += 4 + slen;
    }
    if (bsize >= 0 && bsize < 12 + xlen + 8) return -1;    // no room for the header and the CRC / ISIZE trailer
    return bsize;
}

// zs: a raw-deflate inflater of the calling thread (inflateInit2(-15) once, reset per block)
bool inflate_block(z_stream &zs, const uint8_t *blk, int bsize, uint8_t *dst, int *dlen)
{
    const int xlen = rd_u16(blk + 10);
    const uint8_t *cdata = blk + 12 + xlen;
    const int clen = bsize - 12 - xlen - 8;
    const uint32_t isize = rd_u32(blk + bsize - 4);
    if (clen < 0 || isize > 65536) return false;
    const uint32_t want = rd_u32(blk + bsize - 8);
    // the block decoder of cv_inflate.cpp first (the 8 bytes of trailer behind the stream are its read slack);
    // anything it rejects, or gets past its own checks but not past the CRC, goes to zlib
#ifndef CV_BAM_ZLIB_ONLY                                   /* development: A/B against zlib alone */
    if (cv_inflate_raw(cdata, clen, dst, (int64_t)isize) == (int64_t)isize && cv_crc32_ieee(0, dst, (int64_t)isize) == want) {
        *dlen = (int)isize;
        return true;
    }
#endif
    if (inflateReset(&zs) != Z_OK) return false;
    zs.next_in = const_cast<uint8_t *>(cdata); zs.avail_in = (uInt)clen;
    zs.next_out = dst; zs.avail_out = isize;             // never past the block's own place in the stream
    const int rc = inflate(&zs, Z_FINISH);
    if (rc != Z_STREAM_END || zs.total_out != isize) return false;
    if (cv_crc32_ieee(0, dst, (int64_t)isize) != want) return false;
    *dlen = (int)isize;
    return true;
}

// read and inflate up to max_blocks BGZF blocks starting at b->next_coff, append to b->data
int fill(cv_bam *b, int max_blocks)
{
    if (b->eof) return 0;
    if (b->data_pos > 0 && b->data_pos == b->data.size()) { b->data.clear(); b->data_pos = 0; }
    else if (b->data_pos > (1u << 22)) { b->data.drop_front(b->data_pos); b->data_pos = 0; }
    const size_t want = (size_t)max_blocks * 65536 + 65536;
    if (!b->comp.resize(want)) { cv_set_error("bam: out of memory"); return 1; }
    if (fseeko(b->fp, (off_t)b->next_coff, SEEK_SET)) { cv_set_error("bam: seek failed"); return 1; }
    const size_t got = fread(b->comp.data(), 1, want, b->fp);
    if (got == 0) { b->eof = true; return 0; }
    std::vector<size_t> boff;
    std::vector<int> bsz;
    size_t off = 0;
    while ((int)boff.size() < max_blocks && off < got) {
        const int s = bgzf_block_size(b->comp.data() + off, got - off);
        if (s < 0) { cv_set_error("bam: %s is not BGZF-compressed at offset %lld", b->path.c_str(), (long long)(b->next_coff + (int64_t)off)); return 1; }
        if (s == 0 || off + (size_t)s > got) break;
        boff.push_back(off); bsz.push_back(s);
        off += (size_t)s;
    }
    if (boff.empty()) {
        if (got < want) { b->eof = true; return 0; }       // trailing garbage shorter than a block
        cv_set_error("bam: truncated BGZF block");
        return 1;
    }
    const size_t nb = boff.size();
    const size_t base = b->data.size();
    // every block states its inflated size in its trailer: the blocks go straight to their final places
    std::vector<size_t> dst(nb + 1, 0);
    for (size_t i = 0; i < nb; i++) {
        const uint32_t isize = rd_u32(b->comp.data() + boff[i] + (size_t)bsz[i] - 4);
        if (isize > 65536) { cv_set_error("bam: corrupt BGZF block at offset %lld", (long long)(b->next_coff + (int64_t)boff[i])); return 1; }
        dst[i + 1] = dst[i] + isize;
    }
    if (!b->data.resize(base + dst[nb])) { cv_set_error("bam: out of memory"); return 1; }
    std::vector<int> dlen(nb, 0);
    std::vector<char> ok(nb, 0);
    int T = b->threads < 1 ? 1 : b->threads;
    if ((size_t)T > nb) T = (int)nb;
    auto work = [&](int t) {
        const size_t i0 = nb * (size_t)t / (size_t)T, i1 = nb * ((size_t)t + 1) / (size_t)T;
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit2(&zs, -15) != Z_OK) return;          // ok[] stays 0: reported as a corrupt block
        for (size_t i = i0; i < i1; i++)
            ok[i] = inflate_block(zs, b->comp.data() + boff[i], bsz[i], b->data.data() + base + dst[i], &dlen[i]) &&
                    (size_t)dlen[i] == dst[i + 1] - dst[i];
        inflateEnd(&zs);
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (size_t i = 0; i < nb; i++)
        if (!ok[i]) { cv_set_error("bam: corrupt BGZF block at offset %lld", (long long)(b->next_coff + (int64_t)boff[i])); return 1; }
    b->next_coff += (int64_t)off;
    return 0;
}

// make at least n unread bytes available; false at end of file
bool need(cv_bam *b, size_t n, int *err)
{
    while (b->data.size() - b->data_pos < n) {
        if (b->eof) return false;
        if (fill(b, 64 * (b->threads > 1 ? b->threads : 1))) { *err = 1; return false; }
    }
    return true;
}

int read_header(cv_bam *b)
{
    int err = 0;
    b->next_coff = 0; b->data.clear(); b->data_pos = 0; b->eof = false;
    if (!need(b, 12, &err)) { if (!err) cv_set_error("bam: %s is empty or truncated", b->path.c_str()); return 1; }
    const uint8_t *d = b->data.data();
    if (memcmp(d, "BAM\1", 4)) { cv_set_error("bam: %s has no BAM magic", b->path.c_str()); return 1; }
    const int64_t l_text = rd_i32(d + 4);
    if (l_text < 0) { cv_set_error("bam: corrupt header (l_text %lld)", (long long)l_text); return 1; }
    if (!need(b, (size_t)(12 + l_text), &err)) { if (!err) cv_set_error("bam: truncated header"); return 1; }
    d = b->data.data();
    const int64_t n_ref = rd_i32(d + 8 + l_text);
    size_t pos = (size_t)(12 + l_text);
    for (int64_t i = 0; i < n_ref; i++) {
        if (!need(b, pos + 4 - b->data_pos, &err)) { if (!err) cv_set_error("bam: truncated reference list"); return 1; }
        const int64_t l_name = rd_i32(b->data.data() + pos);
        if (l_name < 1 || l_name > (1 << 20)) { cv_set_error("bam: corrupt reference list (l_name %lld)", (long long)l_name); return 1; }
        if (!need(b, pos + 8 + (size_t)l_name - b->data_pos, &err)) { if (!err) cv_set_error("bam: truncated reference list"); return 1; }
        const uint8_t *q = b->data.data() + pos + 4;
        ref_t r;
        r.name.assign((const char *)q, (size_t)(l_name > 0 ? l_name - 1 : 0));
        r.len = rd_i32(q + l_name);
        b->refs.push_back(r);
        pos += 8 + (size_t)l_name;
    }
    // virtual offset of the first record: the header may end inside a block; views without an index start from
    // the beginning of the file and skip `pos` bytes
    b->first_record_voff = (int64_t)pos;
    return 0;
}

void load_index(cv_bam *b)
{
    std::string ip = b->path + ".bai";
    FILE *f = fopen(ip.c_str(), "rb");
    if (!f && b->path.size() > 4 && b->path.substr(b->path.size() - 4) == ".bam") {
        ip = b->path.substr(0, b->path.size() - 4) + ".bai";
        f = fopen(ip.c_str(), "rb");
    }
    if (!f) return;
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    if (buf.size() < 8 || memcmp(buf.data(), "BAI\1", 4)) return;
    size_t pos = 8;
    const int64_t n_ref = rd_i32(buf.data() + 4);
    std::vector<std::vector<uint64_t>> lin;
    // (every count is checked against the bytes that are left before it sizes anything: a damaged index is ignored)
    for (int64_t r = 0; r < n_ref; r++) {
        if (pos + 4 > buf.size()) return;
        const int64_t n_bin = rd_i32(buf.data() + pos); pos += 4;
        if (n_bin < 0) return;
        for (int64_t k = 0; k < n_bin; k++) {
            if (pos + 8 > buf.size()) return;
            const int64_t n_chunk = rd_i32(buf.data() + pos + 4);
            if (n_chunk < 0 || (uint64_t)n_chunk > (buf.size() - pos - 8) / 16) return;
            pos += 8 + (size_t)n_chunk * 16;
        }
        if (pos + 4 > buf.size()) return;
        const int64_t n_intv = rd_i32(buf.data() + pos); pos += 4;
        if (n_intv < 0 || (uint64_t)n_intv > (buf.size() - pos) / 8) return;
        std::vector<uint64_t> v((size_t)n_intv);
        for (int64_t k = 0; k < n_intv; k++) v[(size_t)k] = rd_u64(buf.data() + pos + (size_t)k * 8);
        pos += (size_t)n_intv * 8;
        lin.push_back(v);
    }
    if ((int64_t)lin.size() == n_ref && n_ref == (int64_t)b->refs.size()) { b->linear = lin; b->has_index = true; }
}

inline char *put_int(char *w, int64_t v)
{
    if (v < 0) { *w++ = '-'; v = -v; }
    char t[24]; int n = 0;
    do { t[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *w++ = t[--n];
    return w;
}

}  // namespace

extern "C" int cv_bam_open(const char *path, int threads, cv_bam **out)
{
    if (!path || !out) { cv_set_error("cv_bam_open: null argument"); return 1; }
    cv_bam *b = new cv_bam();
    b->path = path;
    b->threads = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    b->fp = fopen(path, "rb");
    if (!b->fp) { cv_set_error("cv_bam_open: cannot open %s", path); delete b; return 1; }
    if (read_header(b)) { fclose(b->fp); delete b; return 1; }
    load_index(b);
    *out = b;
    return 0;
}

extern "C" void cv_bam_close(cv_bam *b)
{
    if (!b) return;
    if (b->fp) fclose(b->fp);
    delete b;
}

extern "C" int cv_bam_nref(const cv_bam *b) { return b ? (int)b->refs.size() : 0; }

extern "C" int cv_bam_ref(const cv_bam *b, int i, const char **name, int64_t *len)
{
    if (!b || i < 0 || i >= (int)b->refs.size()) { cv_set_error("cv_bam_ref: bad index"); return 1; }
    if (name) *name = b->refs[(size_t)i].name.c_str();
    if (len) *len = b->refs[(size_t)i].len;
    return 0;
}

extern "C" int cv_bam_has_index(const cv_bam *b) { return b && b->has_index ? 1 : 0; }

// The CIGAR words of a record (rec at refID, its int32 block_size right before it): the inline ones, or the
// CG:B,I array when the inline CIGAR is the long-read placeholder.  0 ok, 1 = placeholder without a usable tag.
extern "C" int cv_bam_record_cigar(const uint8_t *rec, const uint8_t **ops, int64_t *n)
{
    const int64_t bs = rd_i32(rec - 4);
    const int l_name = rec[8];
    const int n_cig = rd_u16(rec + 12);
    const int64_t l_seq = rd_i32(rec + 16);
    const uint8_t *cig = rec + 32 + l_name;
    *ops = cig; *n = n_cig;
    if (!(n_cig == 2 && (rd_u32(cig) & 15) == 4 && (int64_t)(rd_u32(cig) >> 4) == l_seq && (rd_u32(cig + 4) & 15) == 3))
        return 0;
    const uint8_t *a = cig + 8 + (l_seq + 1) / 2 + l_seq, *const end = rec + bs;
    while (a + 3 <= end) {
        const char t0 = (char)a[0], t1 = (char)a[1], ty = (char)a[2];
        a += 3;
        int64_t sz;
        switch (ty) {
        case 'A': case 'c': case 'C': sz = 1; break;
        case 's': case 'S': sz = 2; break;
        case 'i': case 'I': case 'f': sz = 4; break;
        case 'Z': case 'H': { const void *z = memchr(a, 0, (size_t)(end - a)); if (!z) return 1; sz = (const uint8_t *)z - a + 1; break; }
        case 'B': {
            if (a + 5 > end) return 1;
            const char sub = (char)a[0];
            const int64_t cnt = rd_i32(a + 1);
            const int64_t es = (sub == 'c' || sub == 'C') ? 1 : (sub == 's' || sub == 'S') ? 2 : (sub == 'i' || sub == 'I' || sub == 'f') ? 4 : 0;
            if (!es || cnt < 0 || a + 5 + cnt * es > end) return 1;
            if (t0 == 'C' && t1 == 'G' && sub == 'I') { *ops = a + 5; *n = cnt; return 0; }
            sz = 5 + cnt * es;
            break;
        }
        default: return 1;
        }
        if (a + sz > end) return 1;
        a += sz;
    }
    return 1;
}

extern "C" int cv_bam_view_begin(cv_bam *b, const char *ref, int64_t beg1, int64_t end1, int exclude_flags, int with_qual)
{
    if (!b || !ref) { cv_set_error("cv_bam_view_begin: null argument"); return 1; }
    b->tid = -1;
    for (size_t i = 0; i < b->refs.size(); i++)
        if (b->refs[i].name == ref) { b->tid = (int)i; break; }
    b->done = false;
    b->exclude = exclude_flags; b->with_qual = with_qual;
    if (b->tid < 0) { b->done = true; return 0; }         // samtools prints nothing for an unknown contig (and warns)
    if (beg1 <= 0 && end1 <= 0) { b->beg0 = 0; b->end0 = (int64_t)1 << 40; }
    else { b->beg0 = beg1 > 0 ? beg1 - 1 : 0; b->end0 = end1 > 0 ? end1 : (int64_t)1 << 40; }
    // where to start: the linear index gives the first record overlapping the 16 kb window of `beg`; records of
    // later windows can only lie further on in a coordinate-sorted file
    uint64_t voff = (uint64_t)b->first_record_voff;       // offset inside the inflated stream from the file start
    bool from_start = true;
    if (b->has_index) {
        const std::vector<uint64_t> &li = b->linear[(size_t)b->tid];
        size_t w = (size_t)(b->beg0 >> 14);
        uint64_t v = 0;
        if (!li.empty()) {
            if (w >= li.size()) w = li.size() - 1;
            v = li[w];
            while (v == 0 && w + 1 < li.size()) v = li[++w];   // empty windows hold 0: the next filled one is a valid start
        }
        if (v) { voff = v; from_start = false; }
        else if (li.empty()) { b->done = true; return 0; }     // no alignment on this contig
    }
    b->data.clear(); b->data_pos = 0; b->eof = false;
    if (from_start) {
        b->next_coff = 0;
        int err = 0;
        if (!need(b, (size_t)voff, &err)) { b->done = true; return err; }
        b->data_pos = (size_t)voff;
    } else {
        b->next_coff = (int64_t)(voff >> 16);
        int err = 0;
        if (!need(b, (size_t)(voff & 0xffff), &err)) { b->done = true; return err; }
        b->data_pos = (size_t)(voff & 0xffff);
    }
    return 0;
}

// The same selection as cv_bam_view_read, without the text: gathers the next run of selected records -- about
// max_bytes of inflated BAM -- and returns their count; *base + (*offs)[i] is record i at its refID field (the
// int32 block_size precedes it), laid out as in the SAM/BAM specification section 4.2.  The pointers stay
// valid until the next call on this handle.  0 with *done = 1 at the end of the view, -1 on error.
extern "C" int64_t cv_bam_view_records(cv_bam *b, int64_t max_bytes, const uint8_t **base, const uint32_t **offs,
                                       int *done)
{
    if (!b || !base || !offs || max_bytes < 65536) { cv_set_error("cv_bam_view_records: bad argument"); return -1; }
    b->rec_offs.clear();
    *base = nullptr; *offs = nullptr;
    if (max_bytes > ((int64_t)1 << 31)) max_bytes = (int64_t)1 << 31;
    while (!b->done) {
        // buffer the window first (fill() may move the data), then walk it
        while (!b->eof && (int64_t)(b->data.size() - b->data_pos) < max_bytes)
            if (fill(b, 64 * (b->threads > 1 ? b->threads : 1))) return -1;
        size_t pos = b->data_pos;
        const size_t lim = b->data.size();
        const uint8_t *d = b->data.data();
        bool partial = false;
        while (true) {
            if (lim - pos < 4) { partial = lim != pos; break; }
            const int64_t bs = rd_i32(d + pos);
            if (bs < 32) { cv_set_error("bam: corrupt record (block_size %lld)", (long long)bs); return -1; }
            if ((int64_t)(lim - pos) < 4 + bs) { partial = true; break; }
            const uint8_t *r = d + pos + 4;
            const int32_t tid = rd_i32(r), rpos = rd_i32(r + 4);
            const int l_name = r[8];
            const int n_cig = rd_u16(r + 12), flag = rd_u16(r + 14);
            const int64_t l_seq = rd_i32(r + 16);
            if (l_seq < 0 || bs > kMaxRecord || 32 + (int64_t)l_name + 4 * (int64_t)n_cig + (l_seq + 1) / 2 + l_seq > bs) {
                cv_set_error("bam: corrupt record layout");
                return -1;
            }
            if (tid < 0 || tid > b->tid || (tid == b->tid && (int64_t)rpos >= b->end0)) { b->done = true; break; }
            const uint8_t *cig = r + 32 + l_name;
            bool take = tid == b->tid && !(flag & b->exclude);
            if (take) {
                if ((int64_t)rpos < b->beg0) {   // starts left of the region: does it reach in?  (else it overlaps anyway)
                    int64_t span = 0;
                    for (int k = 0; k < n_cig; k++) {
                        const uint32_t c = rd_u32(cig + 4 * k);
                        const int op = (int)(c & 15);
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += c >> 4;
                    }
                    if (span < 1) span = 1;
                    if ((int64_t)rpos + span <= b->beg0) take = false;
                }
                if (take && n_cig == 2) {        // the long-read placeholder must come with its CG tag
                    const uint8_t *ops; int64_t nops;
                    if (cv_bam_record_cigar(r, &ops, &nops)) { cv_set_error("bam: placeholder CIGAR without a CG:B,I tag"); return -1; }
                }
            }
            if (take) b->rec_offs.push_back((uint32_t)(pos + 4));
            pos += (size_t)(4 + bs);
            if ((int64_t)(pos - b->data_pos) >= max_bytes) break;
        }
        const bool advanced = pos != b->data_pos;
        b->data_pos = pos;
        if (b->done) break;
        if (partial && b->eof) { b->done = true; cv_set_error("bam: truncated record"); return -1; }
        if (!partial && b->eof && pos == lim) { b->done = true; break; }
        if (!b->rec_offs.empty()) break;       // hand out what this window held
        if (partial && !advanced) {            // one record larger than the window: widen it
            max_bytes *= 2;
            if (max_bytes > ((int64_t)1 << 31)) { cv_set_error("bam: record larger than 2 GiB"); return -1; }
        }
    }
    if (done) *done = b->done ? 1 : 0;
    *base = b->data.data();
    *offs = b->rec_offs.data();
    return (int64_t)b->rec_offs.size();
}

// Appends whole SAM lines to buf (cap bytes); returns the bytes written (0 with *done = 1 at the end of the view), -1 on error.
extern "C" int64_t cv_bam_view_read(cv_bam *b, char *buf, int64_t cap, int *done)
{
    if (!b || !buf || cap < 1024) { cv_set_error("cv_bam_view_read: bad argument"); return -1; }
    static const char OPS[] = "MIDNSHP=X";
    static const char NT[] = "=ACMGRSVTWYHKDBN";
    char *w = buf, *wend = buf + cap;
    int err = 0;
    while (!b->done) {
        if (!need(b, 4, &err)) { if (err) return -1; b->done = true; break; }
        const int64_t bs = rd_i32(b->data.data() + b->data_pos);
        if (bs < 32) { cv_set_error("bam: corrupt record (block_size %lld)", (long long)bs); return -1; }
        if (!need(b, (size_t)(4 + bs), &err)) { if (!err) cv_set_error("bam: truncated record"); return -1; }
        const uint8_t *r = b->data.data() + b->data_pos + 4;
        const int32_t tid = rd_i32(r), pos = rd_i32(r + 4);
        const int l_name = r[8], mapq = r[9];
        const int n_cig = rd_u16(r + 12), flag = rd_u16(r + 14);
        const int64_t l_seq = rd_i32(r + 16);
        const int32_t ntid = rd_i32(r + 20), npos = rd_i32(r + 24), tlen = rd_i32(r + 28);
        const int64_t fixed = 32 + (int64_t)l_name + 4 * (int64_t)n_cig + (l_seq + 1) / 2 + l_seq;
        if (l_seq < 0 || bs > kMaxRecord || fixed > bs) { cv_set_error("bam: corrupt record layout"); return -1; }
        // sorted file: stop at the first record past the region; unmapped reads (tid -1) sit at the end
        if (tid < 0 || tid > b->tid || (tid == b->tid && (int64_t)pos >= b->end0)) { b->done = true; break; }
        const uint8_t *cig = r + 32 + l_name;
        bool take = tid == b->tid && !(flag & b->exclude);
        if (take) {
            int64_t span = 0;
            for (int k = 0; k < n_cig; k++) {
                const uint32_t c = rd_u32(cig + 4 * k);
                const int op = (int)(c & 15);
                if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += c >> 4;
            }
            if (span < 1) span = 1;
            if ((int64_t)pos + span <= b->beg0) take = false;
        }
        const uint8_t *rcig = cig;
        int64_t rn = n_cig;
        if (take && cv_bam_record_cigar(r, &rcig, &rn)) { cv_set_error("bam: placeholder CIGAR without a CG:B,I tag"); return -1; }
        if (take) {
            const int64_t mate_name = ntid >= 0 && (size_t)ntid < b->refs.size() ? (int64_t)b->refs[(size_t)ntid].name.size() : 1;
            const int64_t worst = l_name + 16 + (int64_t)b->refs[(size_t)tid].name.size() + mate_name + 12 * 6 + 11 * rn + 2 * l_seq + 16;
            if (wend - w < worst) {
                if (w == buf) { cv_set_error("cv_bam_view_read: buffer of %lld bytes is too small for one record", (long long)cap); return -1; }
                break;                                  // the caller comes back for this record
            }
            memcpy(w, r + 32, (size_t)(l_name > 0 ? l_name - 1 : 0)); w += l_name > 0 ? l_name - 1 : 0;
            *w++ = '\t'; w = put_int(w, flag);
            *w++ = '\t'; memcpy(w, b->refs[(size_t)tid].name.data(), b->refs[(size_t)tid].name.size()); w += b->refs[(size_t)tid].name.size();
            *w++ = '\t'; w = put_int(w, (int64_t)pos + 1);
            *w++ = '\t'; w = put_int(w, mapq);
            *w++ = '\t';
            if (rn == 0) *w++ = '*';
            for (int64_t k = 0; k < rn; k++) {
                const uint32_t c = rd_u32(rcig + 4 * k);
                w = put_int(w, c >> 4);
                *w++ = (c & 15) < 9 ? OPS[c & 15] : '?';
            }
            *w++ = '\t';
            if (ntid < 0) *w++ = '*';
            else if (ntid == tid) *w++ = '=';
            else if ((size_t)ntid < b->refs.size()) { memcpy(w, b->refs[(size_t)ntid].name.data(), b->refs[(size_t)ntid].name.size()); w += b->refs[(size_t)ntid].name.size(); }
            else *w++ = '*';
            *w++ = '\t'; w = put_int(w, (int64_t)npos + 1);
            *w++ = '\t'; w = put_int(w, tlen);
            *w++ = '\t';
            const uint8_t *sq = cig + 4 * n_cig;
            if (l_seq == 0) *w++ = '*';
            for (int64_t k = 0; k < l_seq; k++) *w++ = NT[(sq[k >> 1] >> ((~k & 1) << 2)) & 15];
            *w++ = '\t';
            const uint8_t *ql = sq + (l_seq + 1) / 2;
            if (!b->with_qual || l_seq == 0 || ql[0] == 0xff) *w++ = '*';
            else for (int64_t k = 0; k < l_seq; k++) *w++ = (char)(ql[k] + 33);
            *w++ = '\n';
        }
        b->data_pos += (size_t)(4 + bs);
    }
    if (done) *done = b->done ? 1 : 0;
    return w - buf;
}
