/* libvltf_host.so: native TFRecord batch reader (include/vltf_host.h).  Plain C, built with gcc. */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/vltf_host.h"

static __thread char g_err[512];

static void set_err(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* vlh_last_error(void) { return g_err; }

/* ---- CRC-32C: SSE4.2 crc32 instruction when the compiler targets it, else slicing table ---------------- */
#if !defined(__SSE4_2__)
static uint32_t table[8][256];
static int table_ready;

static void init_table(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0);
        table[0][i] = c;
    }
    for (int t = 1; t < 8; ++t)
        for (uint32_t i = 0; i < 256; ++i) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
    table_ready = 1;
}
#endif

uint32_t vlh_crc32c(const void* data, size_t n) {
    const uint8_t* p = (const uint8_t*)data;
    uint32_t crc = 0xFFFFFFFFu;
#if defined(__SSE4_2__)
    uint64_t c = crc;
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        c = __builtin_ia32_crc32di(c, v);
        p += 8;
        n -= 8;
    }
    crc = (uint32_t)c;
    while (n--) crc = __builtin_ia32_crc32qi(crc, *p++);
#else
    if (!table_ready) init_table();
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
              table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) crc = table[0][(crc ^ *p++) & 0xFF] ^ (crc >> 8);
#endif
    return crc ^ 0xFFFFFFFFu;
}

uint32_t vlh_masked_crc32c(const void* data, size_t n) {
    const uint32_t c = vlh_crc32c(data, n);
    return ((c >> 15) | (c << 17)) + 0xA282EAD8u;
}

/* ---- minimal protobuf walk ------------------------------------------------------------------------------ */
static int rd_varint(const uint8_t* b, size_t end, size_t* pos, uint64_t* out) {
    uint64_t r = 0;
    int shift = 0;
    while (*pos < end && shift < 64) {
        const uint8_t c = b[(*pos)++];
        r |= (uint64_t)(c & 0x7F) << shift;
        if (!(c & 0x80)) {
            *out = r;
            return 0;
        }
        shift += 7;
    }
    return -1;
}

/* next field of a message: returns 0 ok, 1 end, -1 malformed.  For wire type 2, (*val, *len) is the payload. */
static int next_field(const uint8_t* b, size_t end, size_t* pos, uint32_t* field, uint32_t* wt, uint64_t* scalar,
                      const uint8_t** val, size_t* len) {
    if (*pos >= end) return 1;
    uint64_t tag;
    if (rd_varint(b, end, pos, &tag)) return -1;
    *field = (uint32_t)(tag >> 3);
    *wt = (uint32_t)(tag & 7);
    if (*wt == 2) {
        uint64_t l;
        if (rd_varint(b, end, pos, &l) || l > (uint64_t)(end - *pos)) return -1;
        *val = b + *pos;
        *len = (size_t)l;
        *pos += (size_t)l;
    } else if (*wt == 0) {
        if (rd_varint(b, end, pos, scalar)) return -1;
    } else if (*wt == 5) {
        if (*pos + 4 > end) return -1;
        *pos += 4;
    } else if (*wt == 1) {
        if (*pos + 8 > end) return -1;
        *pos += 8;
    } else {
        return -1;
    }
    return 0;
}

/* int64 list payload (field 3 of Feature) -> values */
static int int64_list(const uint8_t* feat, size_t flen, int64_t* out, int max, int* count) {
    size_t p = 0;
    uint32_t f, wt;
    uint64_t sc;
    const uint8_t* v;
    size_t l;
    int rc, n = 0;
    while ((rc = next_field(feat, flen, &p, &f, &wt, &sc, &v, &l)) == 0) {
        if (f != 1) continue;
        if (wt == 2) { /* packed */
            size_t q = 0;
            while (q < l) {
                uint64_t x;
                if (rd_varint(v, l, &q, &x)) return -1;
                if (n < max) out[n] = (int64_t)x;
                ++n;
            }
        } else if (wt == 0) {
            if (n < max) out[n] = (int64_t)sc;
            ++n;
        }
    }
    *count = n;
    return rc < 0 ? -1 : 0;
}

static int parse_example(const uint8_t* buf, size_t n, uint8_t* image, int64_t image_bytes, int32_t* dims, int64_t* labels,
                         int max_labels, int32_t* label_count) {
    size_t p0 = 0;
    uint32_t f, wt;
    uint64_t sc;
    const uint8_t *feats, *entry, *v;
    size_t lf, le, lv;
    int rc, got_img = 0;
    dims[0] = dims[1] = dims[2] = 0;
    *label_count = 0;
    while ((rc = next_field(buf, n, &p0, &f, &wt, &sc, &feats, &lf)) == 0) {
        if (f != 1 || wt != 2) continue; /* Example.features */
        size_t p1 = 0;
        while ((rc = next_field(feats, lf, &p1, &f, &wt, &sc, &entry, &le)) == 0) {
            if (f != 1 || wt != 2) continue; /* map entry */
            const uint8_t *key = NULL, *feat = NULL;
            size_t klen = 0, flen = 0, p2 = 0;
            while ((rc = next_field(entry, le, &p2, &f, &wt, &sc, &v, &lv)) == 0) {
                if (f == 1 && wt == 2) { key = v; klen = lv; }
                if (f == 2 && wt == 2) { feat = v; flen = lv; }
            }
            if (rc < 0 || !key || !feat) return -4;
            /* Feature: oneof bytes_list=1 / float_list=2 / int64_list=3 */
            size_t p3 = 0;
            while ((rc = next_field(feat, flen, &p3, &f, &wt, &sc, &v, &lv)) == 0) {
                if (wt != 2) continue;
                if (f == 1 && klen == 9 && !memcmp(key, "image_raw", 9)) {
                    size_t p4 = 0;
                    const uint8_t* img;
                    size_t li;
                    uint32_t f4, w4;
                    if (next_field(v, lv, &p4, &f4, &w4, &sc, &img, &li) != 0 || f4 != 1 || w4 != 2) return -4;
                    if ((int64_t)li != image_bytes) {
                        set_err("image_raw has %zu bytes, expected %lld", li, (long long)image_bytes);
                        return -4;
                    }
                    memcpy(image, img, li);
                    got_img = 1;
                } else if (f == 3) {
                    int64_t tmp[1] = {0};
                    int cnt = 0;
                    if (klen == 5 && !memcmp(key, "label", 5)) {
                        if (int64_list(v, lv, labels, max_labels, &cnt)) return -4;
                        *label_count = cnt;
                    } else if (klen == 6 && !memcmp(key, "height", 6)) {
                        if (int64_list(v, lv, tmp, 1, &cnt) || cnt < 1) return -4;
                        dims[0] = (int32_t)tmp[0];
                    } else if (klen == 5 && !memcmp(key, "width", 5)) {
                        if (int64_list(v, lv, tmp, 1, &cnt) || cnt < 1) return -4;
                        dims[1] = (int32_t)tmp[0];
                    } else if (klen == 5 && !memcmp(key, "depth", 5)) {
                        if (int64_list(v, lv, tmp, 1, &cnt) || cnt < 1) return -4;
                        dims[2] = (int32_t)tmp[0];
                    }
                }
            }
            if (rc < 0) return -4;
        }
        if (rc < 0) return -4;
    }
    if (rc < 0 || !got_img) {
        if (!g_err[0]) set_err("malformed tf.train.Example");
        return -4;
    }
    return 0;
}

static int read_full(int fd, void* dst, size_t n, int64_t off) {
    uint8_t* p = (uint8_t*)dst;
    while (n) {
        const ssize_t r = pread(fd, p, n, off);
        if (r < 0) {
            if (errno == EINTR) continue;
            return -2;
        }
        if (r == 0) return -1;
        p += r;
        n -= (size_t)r;
        off += r;
    }
    return 0;
}

/* A record (12-byte header at `offset`, `len` payload bytes, 4-byte CRC) must lie inside the file: lengths come from the file and are
 * not trusted (a huge value would wrap `len + 4` and the offset arithmetic). */
static int record_fits(uint64_t len, int64_t offset, int64_t file_size) {
    return offset >= 0 && file_size - offset >= 16 && len <= (uint64_t)(file_size - offset - 16);
}

int64_t vlh_read_frames(const char* path, int64_t offset, int count, int verify_crc, uint8_t* images, int64_t image_bytes,
                        int32_t* dims, int64_t* labels, int max_labels, int32_t* label_counts, int32_t* records_read) {
    g_err[0] = 0;
    if (records_read) *records_read = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_err("cannot open %s: %s", path, strerror(errno));
        return -2;
    }
    struct stat st0;
    if (fstat(fd, &st0)) {
        set_err("fstat %s: %s", path, strerror(errno));
        close(fd);
        return -2;
    }
    size_t cap = (size_t)image_bytes + 4096;
    uint8_t* buf = (uint8_t*)malloc(cap);
    int64_t rc = 0;
    for (int i = 0; i < count && buf; ++i) {
        uint8_t hdr[12];
        int r = read_full(fd, hdr, 12, offset);
        if (r) { rc = r; break; }
        uint64_t len;
        uint32_t c;
        memcpy(&len, hdr, 8);
        memcpy(&c, hdr + 8, 4);
        if (verify_crc && c != vlh_masked_crc32c(hdr, 8)) { set_err("record %d: corrupted length CRC", i); rc = -3; break; }
        if (!record_fits(len, offset, (int64_t)st0.st_size)) { set_err("record %d: length %llu runs past the end of the file", i, (unsigned long long)len); rc = -1; break; }
        if (len + 4 > cap) {
            cap = (size_t)len + 4;
            uint8_t* nb = (uint8_t*)realloc(buf, cap);
            if (!nb) { free(buf); buf = NULL; break; }
            buf = nb;
        }
        r = read_full(fd, buf, (size_t)len + 4, offset + 12);
        if (r) { rc = r; break; }
        memcpy(&c, buf + len, 4);
        if (verify_crc && c != vlh_masked_crc32c(buf, (size_t)len)) { set_err("record %d: corrupted payload CRC", i); rc = -3; break; }
        const int pr = parse_example(buf, (size_t)len, images + (int64_t)i * image_bytes, image_bytes, dims + 3 * i,
                                     labels + (int64_t)i * max_labels, max_labels, label_counts + i);
        if (pr) { rc = pr; break; }
        offset += 12 + (int64_t)len + 4;
        if (records_read) *records_read = i + 1;
    }
    if (!buf) {
        set_err("out of memory");
        rc = -2;
    }
    free(buf);
    close(fd);
    if (rc == -1 && !g_err[0]) set_err("end of file before %d records", count);
    if (rc == -2 && !g_err[0]) set_err("I/O error: %s", strerror(errno));
    return rc < 0 ? rc : offset;
}

/* ---- the same batch read on several threads --------------------------------------------------------------------------
 * A 64-clip batch is 1024 records = 236 MB; one thread needs ~45 ms for it (pread into a bounce buffer + CRC-32C + copy of the
 * image bytes), longer than the 37 ms the GPU takes to train on it.  Record boundaries are only known by walking the length
 * headers, so the walk (12 bytes per record) stays serial and the payload work -- everything else -- is split over threads. */
typedef struct {
    int fd, verify_crc, first, last, max_labels; /* records [first, last) */
    const int64_t* offs;                          /* payload offset of each record */
    const uint64_t* lens;
    uint8_t* images;
    int64_t image_bytes;
    int32_t* dims;
    int64_t* labels;
    int32_t* label_counts;
    int rc, bad, created;                         /* first failure of this slice: code and record index; thread was started */
    char err[256];
} mt_slice;

static void* mt_worker(void* arg) {
    mt_slice* s = (mt_slice*)arg;
    size_t cap = (size_t)s->image_bytes + 4096;
    uint8_t* buf = (uint8_t*)malloc(cap);
    s->rc = 0;
    s->bad = -1;
    g_err[0] = 0;
    for (int i = s->first; i < s->last; ++i) {
        const uint64_t len = s->lens[i];
        if (!buf || len + 4 > cap) {
            cap = (size_t)len + 4;
            uint8_t* nb = (uint8_t*)realloc(buf, cap);
            if (!nb) { free(buf); buf = NULL; s->rc = -2; s->bad = i; snprintf(s->err, sizeof(s->err), "out of memory"); break; }
            buf = nb;
        }
        int r = read_full(s->fd, buf, (size_t)len + 4, s->offs[i]);
        if (r) { s->rc = r; s->bad = i; snprintf(s->err, sizeof(s->err), "record %d: %s", i, r == -1 ? "end of file" : strerror(errno)); break; }
        uint32_t c;
        memcpy(&c, buf + len, 4);
        if (s->verify_crc && c != vlh_masked_crc32c(buf, (size_t)len)) {
            s->rc = -3; s->bad = i; snprintf(s->err, sizeof(s->err), "record %d: corrupted payload CRC", i); break;
        }
        const int pr = parse_example(buf, (size_t)len, s->images + (int64_t)i * s->image_bytes, s->image_bytes, s->dims + 3 * i,
                                     s->labels + (int64_t)i * s->max_labels, s->max_labels, s->label_counts + i);
        if (pr) { s->rc = pr; s->bad = i; snprintf(s->err, sizeof(s->err), "record %d: %s", i, g_err[0] ? g_err : "malformed tf.train.Example"); break; }
    }
    free(buf);
    return NULL;
}

int64_t vlh_read_frames_mt(const char* path, int64_t offset, int count, int verify_crc, uint8_t* images, int64_t image_bytes,
                           int32_t* dims, int64_t* labels, int max_labels, int32_t* label_counts, int32_t* records_read,
                           int threads) {
    if (threads <= 1 || count < 2 * threads)
        return vlh_read_frames(path, offset, count, verify_crc, images, image_bytes, dims, labels, max_labels, label_counts, records_read);
    g_err[0] = 0;
    if (records_read) *records_read = 0;
    if (threads > 64) threads = 64;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_err("cannot open %s: %s", path, strerror(errno));
        return -2;
    }
    struct stat st;
    int64_t* offs = (int64_t*)malloc(sizeof(int64_t) * (size_t)count);
    uint64_t* lens = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)count);
    mt_slice* sl = (mt_slice*)calloc((size_t)threads, sizeof(mt_slice));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)threads);
    int64_t rc = 0;
    int have = 0;            /* records whose header AND payload lie inside the file */
    if (!offs || !lens || !sl || !th || fstat(fd, &st)) {
        set_err("out of memory / fstat: %s", strerror(errno));
        rc = -2;
    }
    /* serial walk of the length headers */
    for (int i = 0; i < count && rc == 0; ++i) {
        uint8_t hdr[12];
        const int r = read_full(fd, hdr, 12, offset);
        if (r) { rc = r; break; }
        uint64_t len;
        uint32_t c;
        memcpy(&len, hdr, 8);
        memcpy(&c, hdr + 8, 4);
        if (verify_crc && c != vlh_masked_crc32c(hdr, 8)) { set_err("record %d: corrupted length CRC", i); rc = -3; break; }
        if (!record_fits(len, offset, (int64_t)st.st_size)) { rc = -1; break; }
        offs[i] = offset + 12;
        lens[i] = len;
        offset += 12 + (int64_t)len + 4;
        have = i + 1;
    }
    /* payloads of the `have` whole records, in parallel (also when the walk stopped early: the caller wants records_read) */
    if (rc != -2 && rc != -3 && have > 0) {
        const int per = (have + threads - 1) / threads;
        for (int t = 0; t < threads; ++t) {
            mt_slice* s = &sl[t];
            s->fd = fd; s->verify_crc = verify_crc; s->max_labels = max_labels;
            s->first = t * per < have ? t * per : have;
            s->last = (t + 1) * per < have ? (t + 1) * per : have;
            s->offs = offs; s->lens = lens; s->images = images; s->image_bytes = image_bytes; s->dims = dims; s->labels = labels;
            s->label_counts = label_counts;
            s->created = 0;
            if (s->first >= s->last) continue;
            if (pthread_create(&th[t], NULL, mt_worker, s) == 0) s->created = 1;
            else mt_worker(s);                                  /* no thread to be had: do the slice here */
        }
        for (int t = 0; t < threads; ++t)
            if (sl[t].created) pthread_join(th[t], NULL);
        /* the failure with the lowest record index wins, like the serial reader */
        int first_bad = have;
        for (int t = 0; t < threads; ++t)
            if (sl[t].first < sl[t].last && sl[t].rc && sl[t].bad >= 0 && sl[t].bad < first_bad) {
                first_bad = sl[t].bad;
                rc = sl[t].rc;
                set_err("%s", sl[t].err);
            }
        if (first_bad < have) have = first_bad;
    }
    if (records_read) *records_read = have;
    free(offs);
    free(lens);
    free(sl);
    free(th);
    close(fd);
    if (rc == -1 && !g_err[0]) set_err("end of file before %d records", count);
    if (rc == -2 && !g_err[0]) set_err("I/O error: %s", strerror(errno));
    return rc < 0 ? rc : offset;
}

int64_t vlh_skip_records(const char* path, int64_t offset, int64_t count, int verify_crc) {
    g_err[0] = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) {
        set_err("cannot open %s: %s", path, strerror(errno));
        return -2;
    }
    int64_t rc = 0;
    struct stat st;
    if (fstat(fd, &st)) {
        set_err("fstat %s: %s", path, strerror(errno));
        close(fd);
        return -2;
    }
    for (int64_t i = 0; i < count; ++i) {
        uint8_t hdr[12];
        const int r = read_full(fd, hdr, 12, offset);
        if (r) { rc = r; break; }
        uint64_t len;
        uint32_t c;
        memcpy(&len, hdr, 8);
        memcpy(&c, hdr + 8, 4);
        if (verify_crc && c != vlh_masked_crc32c(hdr, 8)) { set_err("record %lld: corrupted length CRC", (long long)i); rc = -3; break; }
        if (!record_fits(len, offset, (int64_t)st.st_size)) { set_err("record %lld: length %llu runs past the end of the file", (long long)i, (unsigned long long)len); rc = -1; break; }
        offset += 12 + (int64_t)len + 4;
    }
    close(fd);
    if (rc == -1 && !g_err[0]) set_err("end of file before %lld records", (long long)count);
    return rc < 0 ? rc : offset;
}
