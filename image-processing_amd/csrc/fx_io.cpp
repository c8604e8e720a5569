// libfx host I/O: many small files in and out on native threads.
//
// The reference loads one .npy cloud per pool task with np.load
// (src/features/extract_3d_pca_features.py:48-56) and writes one descriptor per
// cloud with np.save (:148-158).  In Python that is ~150 us of interpreter time
// per file, which no thread pool hides; here a batch of files is read straight
// into the caller's (page-locked) buffer and a batch of descriptors is written
// by native threads.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "../../include/fx.h"

void fx_set_error(const char *fmt, ...);

namespace {

struct NpyHeader {
    int itemsize = 0;      // 4 (<f4) or 8 (<f8)
    int64_t rows = 0, cols = 0;
    int64_t data_offset = 0;
};

// Parses the header of an .npy file (format 1.0 - 3.0) that holds a 2-D
// little-endian C-order float32/float64 array.  Returns FX_IO_OK or a status.
int parse_npy_header(const unsigned char *buf, size_t len, NpyHeader *h)
{
    if (len < 12 || memcmp(buf, "\x93NUMPY", 6) != 0)
        return FX_IO_EFORMAT;
    const int major = buf[6];
    size_t hlen, off;
    if (major == 1) {
        hlen = (size_t)buf[8] | ((size_t)buf[9] << 8);
        off = 10;
    } else if (major == 2 || major == 3) {
        hlen = (size_t)buf[8] | ((size_t)buf[9] << 8) | ((size_t)buf[10] << 16) | ((size_t)buf[11] << 24);
        off = 12;
    } else {
        return FX_IO_EFORMAT;
    }
    if (off + hlen > len)
        return FX_IO_EFORMAT;
    const char *t = (const char *)buf + off;
    const std::string_view text(t, hlen);
    // Only the header numpy.save writes is taken, character for character:
    //   {'descr': '<f4', 'fortran_order': False, 'shape': (R, C), } + spaces + '\n'
    // A header in any other form (another writer's key order or spacing, or damaged bytes that np.load would refuse)
    // leaves the file to np.load, which then decides as the reference's loader does.
    size_t p = 0;
    auto lit = [&](const char *s) -> bool {
        const size_t n = strlen(s);
        if (text.substr(p, n) != std::string_view(s, n))
            return false;
        p += n;
        return true;
    };
    auto number = [&](int64_t *v) -> bool {
        if (p >= hlen || text[p] < '0' || text[p] > '9' || (text[p] == '0' && p + 1 < hlen && text[p + 1] >= '0' && text[p + 1] <= '9'))
            return false;
        int64_t x = 0;
        int digits = 0;
        while (p < hlen && text[p] >= '0' && text[p] <= '9') {
            if (++digits > 15)
                return false;
            x = x * 10 + (text[p++] - '0');
        }
        *v = x;
        return true;
    };
    if (!lit("{'descr': '"))
        return FX_IO_EFORMAT;
    if (lit("<f4"))
        h->itemsize = 4;
    else if (lit("<f8"))
        h->itemsize = 8;
    else
        return FX_IO_EDTYPE;
    if (!lit("', 'fortran_order': "))
        return FX_IO_EFORMAT;
    if (lit("True"))
        return FX_IO_EDTYPE;
    if (!lit("False, 'shape': ("))
        return FX_IO_EFORMAT;
    int64_t rows = 0, cols = 0;
    if (!number(&rows))
        return text.substr(p, 1) == ")" ? FX_IO_ESHAPE : FX_IO_EFORMAT;   // "()" is a scalar
    if (lit(",)"))
        return FX_IO_ESHAPE;                                                // one dimension
    if (!lit(", ") || !number(&cols))
        return FX_IO_EFORMAT;
    if (lit(", "))
        return FX_IO_ESHAPE;                                                // three and more
    if (!lit("), }"))
        return FX_IO_EFORMAT;
    while (p + 1 < hlen && text[p] == ' ')
        p++;
    if (p + 1 != hlen || text[p] != '\n')
        return FX_IO_EFORMAT;
    h->rows = rows;
    h->cols = cols;
    h->data_offset = (int64_t)(off + hlen);
    return FX_IO_OK;
}

bool read_fully(int fd, void *dst, size_t n, off_t off)
{
    unsigned char *p = (unsigned char *)dst;
    while (n) {
        const ssize_t r = pread(fd, p, n, off);
        if (r <= 0)
            return false;
        p += r;
        off += r;
        n -= (size_t)r;
    }
    return true;
}

int read_one_points(const char *path, int64_t want_rows, float *out, int64_t *rows_found)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        return FX_IO_EOPEN;
    unsigned char head[4096];
    const ssize_t got = pread(fd, head, sizeof(head), 0);
    NpyHeader h;
    int st = got > 0 ? parse_npy_header(head, (size_t)got, &h) : FX_IO_EFORMAT;
    if (st == FX_IO_OK) {
        *rows_found = h.rows;
        if (h.cols < 3)
            st = FX_IO_ESHAPE;
        else if (h.rows != want_rows)
            st = FX_IO_EROWS;
    }
    if (st == FX_IO_OK) {
        // the header's shape against the file's size, before anything is sized by it
        struct stat sb;
        const unsigned __int128 need = (unsigned __int128)(uint64_t)h.rows * (uint64_t)h.cols * (uint64_t)h.itemsize;
        if (fstat(fd, &sb) != 0 || need > (unsigned __int128)(sb.st_size > h.data_offset ? sb.st_size - h.data_offset : 0))
            st = FX_IO_EREAD;
    }
    if (st == FX_IO_OK) {
        if (h.itemsize == 4 && h.cols == 3) {
            if (!read_fully(fd, out, (size_t)want_rows * 3 * sizeof(float), (off_t)h.data_offset))
                st = FX_IO_EREAD;
        } else {
            const size_t bytes = (size_t)h.rows * (size_t)h.cols * (size_t)h.itemsize;
            std::vector<unsigned char> tmp(bytes);
            if (!read_fully(fd, tmp.data(), bytes, (off_t)h.data_offset)) {
                st = FX_IO_EREAD;
            } else if (h.itemsize == 4) {
                const float *s = (const float *)tmp.data();
                for (int64_t r = 0; r < h.rows; r++)
                    for (int c = 0; c < 3; c++)
                        out[r * 3 + c] = s[r * h.cols + c];
            } else {
                // float64 samples are taken over only when float32 holds every one of them exactly (the reference
                // evaluates the float64 values); other files are left to the caller, who keeps them wide
                const double *s = (const double *)tmp.data();
                for (int64_t r = 0; r < h.rows && st == FX_IO_OK; r++)
                    for (int c = 0; c < 3; c++) {
                        const double v = s[r * h.cols + c];
                        const float f = (float)v;
                        if (!((double)f == v) && v == v)   // (NaN stays NaN either way)
                            st = FX_IO_EWIDE;
                        out[r * 3 + c] = f;
                    }
            }
        }
    }
    close(fd);
    return st;
}

int write_one_row(const char *path, const float *row, int64_t len)
{
    // .npy format 1.0, 1-D little-endian float32, header padded so that the
    // data starts at a multiple of 64 bytes (what np.save writes)
    char dict[128];
    const int dn = snprintf(dict, sizeof(dict), "{'descr': '<f4', 'fortran_order': False, 'shape': (%lld,), }",
                            (long long)len);
    size_t hlen = (size_t)dn + 1;  // + '\n'
    const size_t pad = (64 - (10 + hlen) % 64) % 64;
    hlen += pad;
    std::vector<unsigned char> buf(10 + hlen + (size_t)len * sizeof(float));
    memcpy(buf.data(), "\x93NUMPY\x01\x00", 8);
    buf[8] = (unsigned char)(hlen & 255);
    buf[9] = (unsigned char)(hlen >> 8);
    memcpy(buf.data() + 10, dict, (size_t)dn);
    memset(buf.data() + 10 + dn, ' ', pad);
    buf[10 + hlen - 1] = '\n';
    memcpy(buf.data() + 10 + hlen, row, (size_t)len * sizeof(float));
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0)
        return FX_IO_EOPEN;
    const unsigned char *p = buf.data();
    size_t n = buf.size();
    int st = FX_IO_OK;
    while (n) {
        const ssize_t w = write(fd, p, n);
        if (w <= 0) {
            st = FX_IO_EREAD;
            break;
        }
        p += w;
        n -= (size_t)w;
    }
    if (close(fd) != 0 && st == FX_IO_OK)
        st = FX_IO_EREAD;
    return st;
}

// ------------------------------------------------------------------ PNG
// Decoder for the PNG files a renderer writes: 8 bits per sample, colour type
// gray, gray + alpha, RGB or RGBA, not interlaced, no tRNS chunk.  Everything
// else (palette, 16 bit, interlace, transparency key, a damaged file) is
// reported and left to PIL, so that what PIL accepts, rejects or converts
// differently never depends on this code.  Gray conversion as PIL's
// convert("L"): (R * 19595 + G * 38470 + B * 7471 + 0x8000) >> 16, alpha ignored.
inline uint32_t be32(const unsigned char *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Reads and checks a PNG file's container: signature, chunk CRCs, IHDR (size into found_hw, channel count), and
// the IDAT payloads joined into `idat` (the zlib stream).
int parse_one_png(const char *path, int want_h, int want_w, int32_t *found_hw, int *channels_out,
                  std::vector<unsigned char> &idat)
{
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        return FX_IO_EOPEN;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size < 8 + 25 + 12 || sb.st_size > (off_t)(1u << 30)) {
        close(fd);
        return FX_IO_EFORMAT;
    }
    std::vector<unsigned char> file((size_t)sb.st_size);
    const bool ok = read_fully(fd, file.data(), file.size(), 0);
    close(fd);
    if (!ok)
        return FX_IO_EREAD;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (memcmp(file.data(), sig, 8) != 0)
        return FX_IO_EFORMAT;
    size_t pos = 8;
    int width = 0, height = 0, channels = 0;
    bool have_ihdr = false, have_iend = false;
    idat.clear();
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const unsigned char *type = &file[pos + 4];
        if ((size_t)len > file.size() - pos - 12)
            return FX_IO_EFORMAT;
        const unsigned char *data = &file[pos + 8];
        if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, (uInt)(len + 4)) != be32(data + len))
            return FX_IO_EFORMAT;
        if (memcmp(type, "IHDR", 4) == 0) {
            if (len != 13 || have_ihdr)
                return FX_IO_EFORMAT;
            have_ihdr = true;
            width = (int)be32(data);
            height = (int)be32(data + 4);
            const int depth = data[8], ctype = data[9], interlace = data[12];
            if (depth != 8 || interlace != 0 || data[10] != 0 || data[11] != 0)
                return FX_IO_EDTYPE;
            channels = ctype == 0 ? 1 : ctype == 4 ? 2 : ctype == 2 ? 3 : ctype == 6 ? 4 : 0;
            if (!channels)
                return FX_IO_EDTYPE;
            if (width <= 0 || height <= 0)
                return FX_IO_EFORMAT;
            found_hw[0] = height;
            found_hw[1] = width;
            if (width != want_w || height != want_h)
                return FX_IO_EROWS;
        } else if (!have_ihdr) {
            return FX_IO_EFORMAT;
        } else if (memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (memcmp(type, "tRNS", 4) == 0 || memcmp(type, "PLTE", 4) == 0) {
            return FX_IO_EDTYPE;
        } else if (memcmp(type, "IEND", 4) == 0) {
            have_iend = true;
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || !have_iend || idat.empty())
        return FX_IO_EFORMAT;
    *channels_out = channels;
    return FX_IO_OK;
}

int decode_one_png(const char *path, int want_h, int want_w, uint8_t *out, int32_t *found_hw)
{
    std::vector<unsigned char> idat;
    int channels = 0;
    const int st = parse_one_png(path, want_h, want_w, found_hw, &channels, idat);
    if (st != FX_IO_OK)
        return st;
    const int width = want_w, height = want_h;
    const size_t stride = (size_t)width * channels;
    std::vector<unsigned char> raw((size_t)height * (stride + 1));
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size())
        return FX_IO_EFORMAT;
    // undo the per-row filters in place
    const int bpp = channels;
    for (int y = 0; y < height; y++) {
        unsigned char *row = raw.data() + (size_t)y * (stride + 1);
        const int ft = row[0];
        unsigned char *cur = row + 1;
        const unsigned char *up = y ? cur - (stride + 1) : nullptr;
        switch (ft) {
        case 0:
            break;
        case 1:
            for (size_t i = bpp; i < stride; i++)
                cur[i] = (unsigned char)(cur[i] + cur[i - bpp]);
            break;
        case 2:
            if (up)
                for (size_t i = 0; i < stride; i++)
                    cur[i] = (unsigned char)(cur[i] + up[i]);
            break;
        case 3:
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                cur[i] = (unsigned char)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; i++) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0;
                const int c = (up && i >= (size_t)bpp) ? up[i - bpp] : 0;
                cur[i] = (unsigned char)(cur[i] + paeth(a, b, c));
            }
            break;
        default:
            return FX_IO_EFORMAT;
        }
    }
    for (int y = 0; y < height; y++) {
        const unsigned char *cur = raw.data() + (size_t)y * (stride + 1) + 1;
        uint8_t *o = out + (size_t)y * width;
        if (channels <= 2) {
            for (int x = 0; x < width; x++)
                o[x] = cur[(size_t)x * channels];
        } else {
            for (int x = 0; x < width; x++) {
                const unsigned char *p = cur + (size_t)x * channels;
                o[x] = (uint8_t)((p[0] * 19595u + p[1] * 38470u + p[2] * 7471u + 0x8000u) >> 16);
            }
        }
    }
    return FX_IO_OK;
}

template <typename F>
void parallel_for(int64_t n, int n_threads, F fn)
{
    n_threads = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
    if ((int64_t)n_threads > n)
        n_threads = (int)(n > 0 ? n : 1);
    std::atomic<int64_t> next(0);
    auto body = [&]() {
        for (;;) {
            const int64_t i0 = next.fetch_add(16);
            if (i0 >= n)
                return;
            const int64_t i1 = i0 + 16 < n ? i0 + 16 : n;
            for (int64_t i = i0; i < i1; i++)
                fn(i);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++)
        pool.emplace_back(body);
    body();
    for (auto &th : pool)
        th.join();
}

}  // namespace

extern "C" int fx_read_npy_points_f32(const char *const *paths, int64_t n, int64_t rows, float *out,
                                      int32_t *status, int64_t *rows_found, int n_threads)
{
    if (n < 0 || rows < 0 || (n > 0 && (!paths || !out || !status || !rows_found))) {
        fx_set_error("fx_read_npy_points_f32: bad arguments (n = %lld, rows = %lld)", (long long)n, (long long)rows);
        return FX_EINVAL;
    }
    parallel_for(n, n_threads, [&](int64_t i) {
        rows_found[i] = -1;
        status[i] = read_one_points(paths[i], rows, out + i * rows * 3, &rows_found[i]);
    });
    return FX_OK;
}

extern "C" int fx_write_npy_rows_f32(const char *const *paths, int64_t n, const float *data, int64_t row_len,
                                     int32_t *status, int n_threads)
{
    if (n < 0 || row_len < 0 || (n > 0 && (!paths || !data || !status))) {
        fx_set_error("fx_write_npy_rows_f32: bad arguments (n = %lld, row_len = %lld)", (long long)n,
                     (long long)row_len);
        return FX_EINVAL;
    }
    parallel_for(n, n_threads, [&](int64_t i) { status[i] = write_one_row(paths[i], data + i * row_len, row_len); });
    return FX_OK;
}

extern "C" int fx_decode_png_gray_u8(const char *const *paths, int64_t n, int H, int W, uint8_t *out, int32_t *status,
                                     int32_t *found_hw, int n_threads)
{
    if (n < 0 || H < 1 || W < 1 || (n > 0 && (!paths || !out || !status || !found_hw))) {
        fx_set_error("fx_decode_png_gray_u8: bad arguments (n = %lld, %dx%d)", (long long)n, H, W);
        return FX_EINVAL;
    }
    parallel_for(n, n_threads, [&](int64_t i) {
        found_hw[2 * i] = found_hw[2 * i + 1] = -1;
        status[i] = decode_one_png(paths[i], H, W, out + (size_t)i * H * W, found_hw + 2 * i);
    });
    return FX_OK;
}

// Upper bound of the staging bytes fx_png_stage needs for these files (their sizes plus the padding per file).
extern "C" int64_t fx_png_stage_bytes(const char *const *paths, int64_t n, int n_threads)
{
    if (n < 0 || (n > 0 && !paths))
        return -1;
    std::atomic<int64_t> total(0);
    parallel_for(n, n_threads, [&](int64_t i) {
        struct stat sb;
        if (stat(paths[i], &sb) == 0 && sb.st_size > 0)
            total.fetch_add(((int64_t)sb.st_size + 63) & ~(int64_t)15);
        else
            total.fetch_add(64);
    });
    return total.load() + 64;
}

extern "C" int fx_png_stage(const char *const *paths, int64_t n, int H, int W, uint8_t *staging, int64_t capacity,
                            int64_t *offset, int32_t *length, int32_t *channels, int32_t *status, int32_t *found_hw,
                            int64_t *used, int n_threads)
{
    if (n < 0 || H < 1 || W < 1 || capacity < 0 ||
        (n > 0 && (!paths || !staging || !offset || !length || !channels || !status || !found_hw))) {
        fx_set_error("fx_png_stage: bad arguments (n = %lld, %dx%d)", (long long)n, H, W);
        return FX_EINVAL;
    }
    std::atomic<int64_t> at(0);
    parallel_for(n, n_threads, [&](int64_t i) {
        found_hw[2 * i] = found_hw[2 * i + 1] = -1;
        offset[i] = 0;
        length[i] = 0;
        channels[i] = 0;
        std::vector<unsigned char> idat;
        int ch = 0;
        int st = parse_one_png(paths[i], H, W, found_hw + 2 * i, &ch, idat);
        if (st == FX_IO_OK) {
            // the stream starts 16-byte aligned and is followed by at least 16 zero bytes: the device reads words
            const int64_t need = (((int64_t)idat.size() + 15) & ~(int64_t)15) + 16;
            const int64_t o = at.fetch_add(need);
            if (idat.size() > (size_t)0x7fffffff || o + need > capacity) {
                st = FX_IO_ENOSPACE;
            } else {
                memcpy(staging + o, idat.data(), idat.size());
                memset(staging + o + idat.size(), 0, (size_t)(need - (int64_t)idat.size()));
                offset[i] = o;
                length[i] = (int32_t)idat.size();
                channels[i] = ch;
            }
        }
        status[i] = st;
    });
    if (used)
        *used = at.load() < capacity ? at.load() : capacity;
    return FX_OK;
}
