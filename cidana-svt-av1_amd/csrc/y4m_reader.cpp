// y4m_reader.cpp — the y4m header / frame reader of the C ABI (include/svt_hip_dsp.h, SURVEY 8f n4: the y4m -> plane upload
// path).  Plain host C++: no HIP, no device — built into libsvt_hip_dsp.so and, by tests/test_host_sanitizers.py, into
// AddressSanitizer / UBSan / ThreadSanitizer builds of the host-only pieces (with -DSVT_HIP_TEST_HOOKS for the fault hook).
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <system_error>
#include <thread>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include "host_err.h"

using namespace svthost;

#ifdef SVT_HIP_TEST_HOOKS
// reader threads with an index >= this value fail to start (0 = none fail): drives the "no thread to be had" branch
static std::atomic<int> g_fail_threads_from{0};
extern "C" void svt_hip_test_y4m_fail_threads_from(int t) { g_fail_threads_from.store(t); }
#endif

// ---- y4m (Source/App/EncApp/EbAppInputy4m.c) -------------------------------------------------------------------------
namespace {
// copyUntilCharacterOrNewLine (:13-33) as the application's build behaves: EB_STRNCPY(dst, src, count) bounds the copy with
// sizeof(dst) of a char POINTER, so a token of 8 or more characters (or an empty one) is cleared by strncpy_ss
// (EbAppFifo.c:160-283) and compares equal to nothing.  Kept, so that this reader accepts and rejects exactly the files the
// reference's application does ("C420mpeg2" / "C420paldv" are rejected there).
const char* y4m_token(const char* src, char* dst, size_t cap, char chr) {
    const char* s0 = src;
    size_t n = 0;
    while (*src != chr && *src != '\n' && *src != '\0') { src++; n++; }
    if (n == 0 || n >= 8 || n + 1 > cap) dst[0] = '\0';
    else { memcpy(dst, s0, n); dst[n] = '\0'; }
    return src;
}
struct Y4mFmt { const char* name; const char* chroma; uint32_t bd; };
const Y4mFmt kY4mFmt[] = {      // the 'C' tokens of read_y4m_header (:93-190)
    {"420mpeg2", "420", 8}, {"420paldv", "420", 8}, {"420jpeg", "420", 8}, {"420p16", "420", 16}, {"422p16", "422", 16}, {"444p16", "444", 16},
    {"420p14", "420", 14}, {"422p14", "422", 14}, {"444p14", "444", 14}, {"420p12", "420", 12}, {"422p12", "422", 12}, {"444p12", "444", 12},
    {"420p10", "420", 10}, {"422p10", "422", 10}, {"444p10", "444", 10}, {"420p9", "420", 9}, {"422p9", "422", 9}, {"444p9", "444", 9},
    {"420", "420", 8}, {"411", "411", 8}, {"422", "422", 8}, {"444", "444", 8},
    {"mono16", "400", 16}, {"mono12", "400", 12}, {"mono10", "400", 10}, {"mono9", "400", 9}, {"mono", "400", 8}};
}  // namespace

struct svt_hip_y4m {
    FILE* f;
    svt_hip_y4m_info info;
    size_t frame_bytes;
};

extern "C" int svt_hip_y4m_parse_header(const char* line, svt_hip_y4m_info* out) {
    if (!line || !out) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    uint32_t bitdepth = 8, width = 0, height = 0, fr_n = 0, fr_d = 0;
    char chroma[8] = "420", scan = 'p', tok[96];
    uint32_t interlaced = 1;                              // read_y4m_header starts from interlaced = EB_TRUE (:43)
    for (const char* p = line; *p != '\0'; p++) {
        if (*p == 0x20) continue;
        switch (*p++) {
        case 'W': { char* e; width = (uint32_t)strtol(p, &e, 10); p = e; } break;
        case 'H': { char* e; height = (uint32_t)strtol(p, &e, 10); p = e; } break;
        case 'I':
            switch (*p++) {
            case 'p': interlaced = 0; scan = 'p'; break;
            case 't': interlaced = 1; scan = 't'; break;
            case 'b': interlaced = 1; scan = 'b'; break;
            default: return set_err(SVT_HIP_ERR_INVALID, "interlace type not supported");
            }
            break;
        case 'C': {
            p = y4m_token(p, tok, sizeof(tok), 0x20);
            const Y4mFmt* f = nullptr;
            for (const Y4mFmt& c : kY4mFmt)
                if (strcmp(c.name, tok) == 0) { f = &c; break; }
            if (!f) return set_err(SVT_HIP_ERR_INVALID, "chroma format not supported");
            strcpy(chroma, f->chroma);
            bitdepth = f->bd;
        } break;
        case 'F':
            p = y4m_token(p, tok, sizeof(tok), ':');
            fr_n = (uint32_t)strtol(tok, nullptr, 10);
            if (*p != '\0') p++;
            p = y4m_token(p, tok, sizeof(tok), 0x20);
            fr_d = (uint32_t)strtol(tok, nullptr, 10);
            break;
        case 'A':
            p = y4m_token(p, tok, sizeof(tok), ':');
            if (*p != '\0') p++;
            p = y4m_token(p, tok, sizeof(tok), 0x20);
            break;
        default: break;
        }
        if (*p == '\0') break;
    }
    if (width == 0) return set_err(SVT_HIP_ERR_INVALID, "width not found in y4m header");
    if (height == 0) return set_err(SVT_HIP_ERR_INVALID, "height not found in y4m header");
    if (fr_n == 0 || fr_d == 0) return set_err(SVT_HIP_ERR_INVALID, "frame rate not found in y4m header");
    memset(out, 0, sizeof(*out));
    out->width = width; out->height = height; out->fr_n = fr_n; out->fr_d = fr_d;
    out->bit_depth = bitdepth; out->interlaced = interlaced; out->scan_type = scan;
    strcpy(out->chroma, chroma);
    return SVT_HIP_OK;
}

// bytes of one frame's planes in the file: 8-bit samples are bytes, deeper ones 16-bit little endian
extern "C" size_t svt_hip_y4m_frame_bytes(const svt_hip_y4m_info* info) {
    if (!info) return 0;
    const size_t es = info->bit_depth > 8 ? 2 : 1, luma = (size_t)info->width * info->height;
    size_t chroma = 0;
    if (!strcmp(info->chroma, "420")) chroma = 2 * ((size_t)((info->width + 1) >> 1) * ((info->height + 1) >> 1));
    else if (!strcmp(info->chroma, "422")) chroma = 2 * ((size_t)((info->width + 1) >> 1) * info->height);
    else if (!strcmp(info->chroma, "444")) chroma = 2 * luma;
    else if (!strcmp(info->chroma, "411")) chroma = 2 * ((size_t)((info->width + 3) >> 2) * info->height);
    return (luma + chroma) * es;
}

// check_if_y4m (:269-290) + read_y4m_header: SVT_HIP_ERR_INVALID for a file that does not start with "YUV4MPEG2" or whose
// header the reference rejects
extern "C" int svt_hip_y4m_open(const char* path, svt_hip_y4m** out, svt_hip_y4m_info* info) {
    if (!path || !out) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return set_err(SVT_HIP_ERR_INVALID, "cannot open %s", path);
    char sig[10] = {0}, line[80];                        // YFM_HEADER_MAX = 80 (:7): the header line is at most 79 characters
    if (fread(sig, 9, 1, f) != 1 || strcmp(sig, "YUV4MPEG2") != 0) { fclose(f); return set_err(SVT_HIP_ERR_INVALID, "%s is not a YUV4MPEG2 file", path); }
    if (!fgets(line, sizeof(line), f)) { fclose(f); return set_err(SVT_HIP_ERR_INVALID, "%s: no header line", path); }
    svt_hip_y4m_info inf;
    if (int rc = svt_hip_y4m_parse_header(line, &inf)) { fclose(f); return rc; }
    svt_hip_y4m* h = new svt_hip_y4m{f, inf, svt_hip_y4m_frame_bytes(&inf)};
    if (info) *info = inf;
    *out = h;
    return SVT_HIP_OK;
}

// read_y4m_frame_delimiter (:247-266) + the frame's planes.  1 = a frame was read, 0 = end of file, < 0 = error
extern "C" int svt_hip_y4m_read_frame(svt_hip_y4m* h, void* host_dst, size_t capacity) {
    if (!h || !host_dst) return set_err(SVT_HIP_ERR_INVALID, "NULL argument");
    if (capacity < h->frame_bytes) return set_err(SVT_HIP_ERR_INVALID, "buffer of %zu bytes for a frame of %zu", capacity, h->frame_bytes);
    char d[10];
    if (!fgets(d, sizeof(d), h->f)) return 0;
    if (strcmp(d, "FRAME\n") != 0) return set_err(SVT_HIP_ERR_INVALID, "Failed to read proper y4m frame delimeter. Read broken.");
    // a large frame of a regular file: the planes are copied out of the page cache by a few threads (pread on disjoint ranges) - one
    // thread's copy (~20 GB/s) is what bounds the file -> HBM path of a 4K 10-bit clip; pipes and small frames take fread
    const off_t pos = ftello(h->f);
    struct stat st;
    const int fd = fileno(h->f);
    const bool regular = pos >= 0 && fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
    if (regular && h->frame_bytes >= ((size_t)4 << 20) && st.st_size >= pos && (size_t)(st.st_size - pos) >= h->frame_bytes) {
        constexpr int NT = 4;
        const size_t chunk = ((h->frame_bytes + NT - 1) / NT + 4095) & ~(size_t)4095;
        bool ok[NT];
        auto work = [&](int t) {
            const size_t b = (size_t)t * chunk, e = b + chunk < h->frame_bytes ? b + chunk : h->frame_bytes;
            size_t done = b;
            while (done < e) {
                const ssize_t r = pread(fd, (char*)host_dst + done, e - done, pos + (off_t)done);
                if (r <= 0) break;
                done += (size_t)r;
            }
            ok[t] = done >= e;
        };
        std::thread th[NT - 1];
        int started = 0;
        try {
            for (int t = 1; t < NT; t++) {
#ifdef SVT_HIP_TEST_HOOKS
                if (g_fail_threads_from.load() && t >= g_fail_threads_from.load()) throw std::system_error(EAGAIN, std::generic_category());
#endif
                th[t - 1] = std::thread(work, t);
                started = t;
            }
        } catch (...) {}                                   // no thread to be had: this one reads the remaining shares itself
        work(0);
        for (int t = started + 1; t < NT; t++) work(t);
        for (int t = 1; t <= started; t++) th[t - 1].join();
        for (int t = 0; t < NT; t++)
            if (!ok[t]) return set_err(SVT_HIP_ERR_INVALID, "read error in a frame of %zu bytes", h->frame_bytes);
        if (fseeko(h->f, pos + (off_t)h->frame_bytes, SEEK_SET) != 0) return set_err(SVT_HIP_ERR_INVALID, "seek past the frame failed");
        return 1;
    }
    const size_t got = fread(host_dst, 1, h->frame_bytes, h->f);
    if (got != h->frame_bytes) return got == 0 ? 0 : set_err(SVT_HIP_ERR_INVALID, "truncated frame: %zu of %zu bytes", got, h->frame_bytes);
    return 1;
}
extern "C" void svt_hip_y4m_close(svt_hip_y4m* h) {
    if (!h) return;
    if (h->f) fclose(h->f);
    delete h;
}
