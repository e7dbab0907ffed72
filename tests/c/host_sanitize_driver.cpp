// tests/c/host_sanitize_driver.cpp — drives the HOST-ONLY pieces of libsvt_hip_dsp (csrc/host_tables.cpp, csrc/y4m_reader.cpp,
// csrc/host_err.cpp: no HIP, no device) so that they can run under AddressSanitizer + UBSan and, in a second build, under
// ThreadSanitizer (tests/test_host_sanitizers.py compiles this file together with those sources; never on a GPU).
// SURVEY 5 asked for a sanitizer pass over the shim's host code; the reference itself only has a Valgrind CI job.
//
//   headers N       N random y4m header lines (valid and damaged tokens, up to 79 characters) through svt_hip_y4m_parse_header
//   frames DIR      a >= 4 MiB-frame y4m file through the four-thread pread path: whole frames, a truncated last frame, a file
//                   that ends inside the delimiter, the thread-start-failure branch (test hook), and a small-frame file (fread)
//   threads DIR     two handles read concurrently from two threads + the table builders called from four threads (TSAN)
//   avail           the 353 528 (has_top_right, has_bottom_left) argument tuples of tests/golden/bip.npz + svt_hip_intra_neighbor_px
//   tables          svt_hip_build_quantizer (8 / 10 / 12 bit), svt_hip_get_scan for every size x type, svt_hip_ois_candidates
// Exit code 0 = every check held (the sanitizers abort the process on a finding).
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "svt_hip_dsp.h"

extern "C" void svt_hip_test_y4m_fail_threads_from(int t);      // -DSVT_HIP_TEST_HOOKS builds of y4m_reader.cpp

static uint64_t g_rng = 13596;
static uint32_t rnd() { g_rng = g_rng * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(g_rng >> 33); }
static uint32_t rnd_below(uint32_t n) { return n ? rnd() % n : 0; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (%s:%d) last error: %s\n", #c, __FILE__, __LINE__, svt_hip_last_error()); exit(1); } } while (0)

static std::string random_header() {
    static const char* fmts[] = {"420", "420jpeg", "420mpeg2", "420paldv", "420p10", "422p10", "444p12", "mono", "mono16", "411", "420p9", "422",
                                 "444", "420p16", "420p14", "bogus", "", "420jpeg2", "4", "420p10xxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxxx"};
    std::string line = " ";
    const int ntok = 1 + (int)rnd_below(7);
    for (int i = 0; i < ntok; i++) {
        char buf[128];
        switch (rnd_below(10)) {
        case 0: snprintf(buf, sizeof buf, "W%u", rnd_below(5000)); break;
        case 1: snprintf(buf, sizeof buf, "H%u", rnd_below(3000)); break;
        case 2: { uint64_t m = 1; for (uint32_t k = rnd_below(11); k; k--) m *= 10; snprintf(buf, sizeof buf, "F%llu:%u", (unsigned long long)(rnd() % (m ? m : 1)), rnd_below(2000)); } break;
        case 3: snprintf(buf, sizeof buf, "I%c", "ptb?x"[rnd_below(5)]); break;
        case 4: snprintf(buf, sizeof buf, "A%u:%u", rnd_below(300), rnd_below(300)); break;
        case 5: snprintf(buf, sizeof buf, "C%s", fmts[rnd_below(sizeof fmts / sizeof fmts[0])]); break;
        case 6: snprintf(buf, sizeof buf, "X%.*s", (int)rnd_below(13), "YSCSS=420JPEG"); break;
        case 7: buf[0] = 0; break;
        case 8: snprintf(buf, sizeof buf, "F:"); break;
        default: snprintf(buf, sizeof buf, "Q7"); break;
        }
        if (i) line += ' ';
        line += buf;
    }
    if (line.size() > 78) line.resize(78);
    if (rnd_below(8)) line += '\n';                       // mostly newline-terminated, sometimes cut by the fgets buffer
    return line;
}

static int cmd_headers(int n) {
    int ok = 0;
    for (int i = 0; i < n; i++) {
        const std::string l = random_header();
        std::vector<char> exact(l.begin(), l.end());      // heap copy of the exact size + NUL: an over-read trips ASAN
        exact.push_back('\0');
        svt_hip_y4m_info info;
        memset(&info, 0xA5, sizeof info);
        const int rc = svt_hip_y4m_parse_header(exact.data(), &info);
        if (rc == SVT_HIP_OK) {
            ok++;
            CHECK(info.width > 0 && info.height > 0 && info.fr_n > 0 && info.fr_d > 0);
            CHECK(strlen(info.chroma) == 3);
            CHECK(svt_hip_y4m_frame_bytes(&info) > 0);
        } else {
            CHECK(rc == SVT_HIP_ERR_INVALID && svt_hip_last_error()[0] != 0);
        }
    }
    CHECK(svt_hip_y4m_parse_header(nullptr, nullptr) == SVT_HIP_ERR_INVALID);
    printf("headers: %d lines, %d accepted\n", n, ok);
    return 0;
}

static uint8_t frame_byte(int frame, size_t i) { return (uint8_t)((i * 2654435761u + (size_t)frame * 977u) >> 7); }

static std::string write_y4m(const std::string& dir, const char* name, int w, int h, const char* ctoken, int es, int nframes, long cut_tail) {
    const std::string path = dir + "/" + name;
    FILE* f = fopen(path.c_str(), "wb");
    CHECK(f != nullptr);
    fprintf(f, "YUV4MPEG2 W%d H%d F30:1 Ip %s\n", w, h, ctoken);
    const size_t fb = ((size_t)w * h + 2 * (size_t)((w + 1) / 2) * ((h + 1) / 2)) * es;
    std::vector<uint8_t> buf(fb);
    for (int k = 0; k < nframes; k++) {
        for (size_t i = 0; i < fb; i++) buf[i] = frame_byte(k, i);
        fputs("FRAME\n", f);
        const size_t nw = (k == nframes - 1 && cut_tail > 0) ? fb - (size_t)cut_tail : fb;
        CHECK(fwrite(buf.data(), 1, nw, f) == nw);
    }
    if (cut_tail < 0) fputs("FRA", f);                    // the file ends inside a delimiter
    fclose(f);
    return path;
}

static int read_all(const std::string& path, size_t expect_fb, int expect_frames, int expect_last_rc) {
    svt_hip_y4m* h = nullptr;
    svt_hip_y4m_info info;
    CHECK(svt_hip_y4m_open(path.c_str(), &h, &info) == SVT_HIP_OK && h);
    const size_t fb = svt_hip_y4m_frame_bytes(&info);
    CHECK(fb == expect_fb);
    std::vector<uint8_t> buf(fb);                         // exactly one frame: a write past the end trips ASAN
    int frames = 0, rc;
    CHECK(svt_hip_y4m_read_frame(h, buf.data(), fb - 1) == SVT_HIP_ERR_INVALID);       // too small a buffer is refused, nothing read
    while ((rc = svt_hip_y4m_read_frame(h, buf.data(), fb)) == 1) {
        for (size_t i = 0; i < fb; i += 4099) CHECK(buf[i] == frame_byte(frames, i));
        CHECK(buf[fb - 1] == frame_byte(frames, fb - 1));
        frames++;
    }
    CHECK(frames == expect_frames);
    CHECK(expect_last_rc == 0 ? rc == 0 : rc < 0);
    svt_hip_y4m_close(h);
    svt_hip_y4m_close(nullptr);
    return frames;
}

static int cmd_frames(const std::string& dir) {
    // 2048 x 1536 4:2:0 8-bit = 4.5 MiB per frame: the four-thread pread path
    const size_t big = (size_t)2048 * 1536 * 3 / 2;
    const std::string whole = write_y4m(dir, "big_whole.y4m", 2048, 1536, "C420jpeg", 1, 3, 0);
    const std::string trunc = write_y4m(dir, "big_trunc.y4m", 2048, 1536, "C420jpeg", 1, 3, 12345);
    const std::string delim = write_y4m(dir, "big_delim.y4m", 2048, 1536, "C420jpeg", 1, 2, -1);
    // 10-bit: 1920 x 1088 x 2 bytes x 1.5 = 6 MiB
    const std::string hbd = write_y4m(dir, "hbd.y4m", 1920, 1088, "C420p10", 2, 2, 0);
    const std::string small = write_y4m(dir, "small.y4m", 352, 288, "C420jpeg", 1, 5, 0);
    const std::string small_trunc = write_y4m(dir, "small_trunc.y4m", 352, 288, "C420jpeg", 1, 2, 100);
    for (int fail_from = 0; fail_from <= 3; fail_from++) {     // 0: all threads start; 1 / 2 / 3: threads from that index on do not
        svt_hip_test_y4m_fail_threads_from(fail_from);
        read_all(whole, big, 3, 0);
        read_all(trunc, big, 2, -1);                      // the truncated last frame is an error, not a short frame
        read_all(delim, big, 2, -1);
        read_all(hbd, (size_t)1920 * 1088 * 3, 2, 0);
    }
    svt_hip_test_y4m_fail_threads_from(0);
    read_all(small, (size_t)352 * 288 * 3 / 2, 5, 0);
    read_all(small_trunc, (size_t)352 * 288 * 3 / 2, 1, -1);
    svt_hip_y4m* h = nullptr;
    CHECK(svt_hip_y4m_open((dir + "/nope.y4m").c_str(), &h, nullptr) == SVT_HIP_ERR_INVALID && h == nullptr);
    FILE* f = fopen((dir + "/notyuv.y4m").c_str(), "wb"); fputs("RIFFxxxxxxxxxxxx\n", f); fclose(f);
    CHECK(svt_hip_y4m_open((dir + "/notyuv.y4m").c_str(), &h, nullptr) == SVT_HIP_ERR_INVALID && h == nullptr);
    f = fopen((dir + "/sigonly.y4m").c_str(), "wb"); fputs("YUV4MPEG2", f); fclose(f);
    CHECK(svt_hip_y4m_open((dir + "/sigonly.y4m").c_str(), &h, nullptr) == SVT_HIP_ERR_INVALID && h == nullptr);
    printf("frames: pread path (0-3 reader threads refused), truncated frames, broken delimiters, fread path: ok\n");
    return 0;
}

static int tables_once() {
    std::vector<int16_t> t(5 * 256 * 8);
    auto row = [&](int k) { return reinterpret_cast<int16_t(*)[8]>(t.data() + (size_t)k * 256 * 8); };
    for (int bd : {8, 10, 12}) {
        CHECK(svt_hip_build_quantizer(bd, row(0), row(1), row(2), row(3), row(4)) == SVT_HIP_OK);
        for (int qi = 0; qi < 256; qi++) CHECK(row(4)[qi][0] > 0 && row(4)[qi][1] > 0);
    }
    CHECK(svt_hip_build_quantizer(9, row(0), row(1), row(2), row(3), row(4)) == SVT_HIP_ERR_INVALID);
    for (int s = 0; s < SVT_TX_SIZES_ALL; s++)
        for (int ty = 0; ty < SVT_TX_TYPES; ty++) {
            std::vector<int16_t> scan(1024), iscan(1024);
            const int n = svt_hip_get_scan(s, ty, scan.data(), iscan.data());
            if (n < 0) continue;
            CHECK(n >= 16 && n <= 1024);
            std::vector<int16_t> sc(n), isc(n);           // exact-size buffers
            CHECK(svt_hip_get_scan(s, ty, sc.data(), isc.data()) == n);
            for (int i = 0; i < n; i++) CHECK(sc[i] >= 0 && sc[i] < n && isc[sc[i]] == i);
        }
    CHECK(svt_hip_get_scan(-1, 0, nullptr, nullptr) < 0 && svt_hip_get_scan(0, 99, nullptr, nullptr) < 0);
    for (uint32_t bs : {8u, 16u, 32u, 64u})
        for (int tl = 0; tl < 4; tl++)
            for (int ipm = 0; ipm <= 6; ipm++)
                for (int ref = 0; ref < 2; ref++)
                    for (int is16 = 0; is16 < 2; is16++) {
                        uint8_t modes[SVT_HIP_OIS_MAX_CANDIDATES];
                        int8_t deltas[SVT_HIP_OIS_MAX_CANDIDATES];
                        const int n = svt_hip_ois_candidates(bs, tl, ipm, ref, is16, modes, deltas);
                        CHECK(n < 0 || (n >= 1 && n <= SVT_HIP_OIS_MAX_CANDIDATES));
                    }
    return 0;
}

static const int kBW[22] = {4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64, 128, 128, 4, 16, 8, 32, 16, 64};
static const int kBH[22] = {4, 8, 4, 8, 16, 8, 16, 32, 16, 32, 64, 32, 64, 128, 64, 128, 16, 4, 32, 8, 64, 16};
static const int kTW[19] = {4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64};
static const int kTH[19] = {4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16};

static int cmd_avail() {
    // the enumeration of tests/svtlibs.py availability_tuples (what tests/golden/bip.npz stores bits for)
    long total = 0, ones = 0;
    for (int sb_mi : {16, 32})
        for (int bs = 0; bs < 22; bs++) {
            if (kBW[bs] > sb_mi * 4 || kBH[bs] > sb_mi * 4) continue;
            const int bw = kBW[bs] / 4, bh = kBH[bs] / 4;
            std::vector<int> parts = {0, 1, 2, 3, 4, 5, 8, 9};
            if (bs >= 1 && bs < 16 && kBW[bs] <= kBH[bs]) { parts.push_back(6); parts.push_back(7); }
            for (int part : parts)
                for (int r = 0; r < sb_mi; r += bh)
                    for (int c = 0; c < sb_mi; c += bw)
                        for (int tx = 0; tx < 19; tx++) {
                            if (kTW[tx] > kBW[bs] || kTH[tx] > kBH[bs]) continue;
                            for (int ss = 0; ss < 2; ss++) {
                                if (ss && (kBW[bs] < 8 || kBH[bs] < 8)) continue;
                                const int tw = kTW[tx] / 4, th = kTH[tx] / 4;
                                const int bwu = (bw >> ss) > 1 ? (bw >> ss) : 1, bhu = (bh >> ss) > 1 ? (bh >> ss) : 1;
                                int offs[4][2] = {{0, 0}}, no = 1;
                                if (tw < bwu) { offs[no][0] = 0; offs[no][1] = tw; no++; }
                                if (th < bhu) { offs[no][0] = th; offs[no][1] = 0; no++; }
                                if (tw < bwu && th < bhu) { offs[no][0] = th; offs[no][1] = tw; no++; }
                                for (int k = 0; k < no; k++) {
                                    const int tr = svt_hip_intra_has_top_right(sb_mi, bs, 64 + r, 96 + c, 1, 1, part, tx, offs[k][0], offs[k][1], ss, ss);
                                    const int bl = svt_hip_intra_has_bottom_left(sb_mi, bs, 64 + r, 96 + c, 1, 1, part, tx, offs[k][0], offs[k][1], ss, ss);
                                    CHECK((tr == 0 || tr == 1) && (bl == 0 || bl == 1));
                                    total++; ones += tr + bl;
                                }
                            }
                        }
        }
    CHECK(total == 353528);
    // svt_hip_intra_neighbor_px on random positions, valid and invalid
    long okc = 0;
    for (int i = 0; i < 200000; i++) {
        svt_hip_intra_pos p;
        memset(&p, 0, sizeof p);
        p.is_16bit = rnd_below(2); p.sb_size_mi = rnd_below(8) ? 16 : (rnd_below(2) ? 32 : 24);
        p.mi_rows = 1 + rnd_below(300); p.mi_cols = 1 + rnd_below(500);
        p.tile_mi_row_start = 0; p.tile_mi_row_end = p.mi_rows; p.tile_mi_col_start = 0; p.tile_mi_col_end = p.mi_cols;
        p.partition = (int)rnd_below(11) - (rnd_below(50) == 0); p.bsize = (int)rnd_below(23) - (rnd_below(50) == 0);
        p.tx_size = (int)rnd_below(20); p.plane = (int)rnd_below(3);
        p.bl_org_x_pict = (int)rnd_below(2000) & ~3; p.bl_org_y_pict = (int)rnd_below(1200) & ~3;
        p.col_off = (int)rnd_below(4); p.row_off = (int)rnd_below(4); p.wpx = 4 << rnd_below(5); p.hpx = 4 << rnd_below(5);
        svt_hip_intra_blk b;
        memset(&b, 0, sizeof b);
        const int rc = svt_hip_intra_neighbor_px(&p, &b);
        CHECK(rc == SVT_HIP_OK || rc == SVT_HIP_ERR_INVALID);
        okc += rc == SVT_HIP_OK;
    }
    CHECK(svt_hip_intra_neighbor_px(nullptr, nullptr) == SVT_HIP_ERR_INVALID);
    printf("avail: %ld tuples (%ld bits set), neighbor_px: %ld of 200000 random positions valid\n", total, ones, okc);
    return 0;
}

static int cmd_threads(const std::string& dir) {
    const std::string a = write_y4m(dir, "ta.y4m", 2048, 1536, "C420jpeg", 1, 3, 0);
    const std::string b = write_y4m(dir, "tb.y4m", 1920, 1088, "C420p10", 2, 2, 0);
    std::vector<std::thread> th;
    th.emplace_back([&] { read_all(a, (size_t)2048 * 1536 * 3 / 2, 3, 0); });
    th.emplace_back([&] { read_all(b, (size_t)1920 * 1088 * 3, 2, 0); });
    th.emplace_back([&] { read_all(a, (size_t)2048 * 1536 * 3 / 2, 3, 0); });      // the same file through a second handle
    for (int i = 0; i < 4; i++) th.emplace_back([] { tables_once(); });
    th.emplace_back([] {                                      // the error text is per thread: concurrent failures do not share it
        for (int i = 0; i < 2000; i++) {
            svt_hip_y4m_info info;
            (void)svt_hip_y4m_parse_header(random_header().c_str(), &info);
        }
    });
    for (auto& t : th) t.join();
    printf("threads: 3 concurrent readers (12 pread workers), 4 table builders, 1 header parser: ok\n");
    return 0;
}

int main(int argc, char** argv) {
    const std::string cmd = argc > 1 ? argv[1] : "";
    if (cmd == "headers") return cmd_headers(argc > 2 ? atoi(argv[2]) : 2000);
    if (cmd == "frames" && argc > 2) return cmd_frames(argv[2]);
    if (cmd == "threads" && argc > 2) return cmd_threads(argv[2]);
    if (cmd == "avail") return cmd_avail();
    if (cmd == "tables") { tables_once(); printf("tables: ok\n"); return 0; }
    fprintf(stderr, "usage: %s headers N | frames DIR | threads DIR | avail | tables\n", argv[0]);
    return 2;
}
