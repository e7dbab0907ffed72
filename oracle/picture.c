/*
 * oracle/picture.c — CPU restatement of the picture-input side of the path ("the y4m -> plane upload path").
 * TEST INFRASTRUCTURE ONLY (see oracle/Makefile): the product never links or calls this file.
 * Pinned to the reference's own code through oracle/_ref (tests/test_oracle_vs_ref.py, fixtures tests/golden/picture.npz):
 *   svt_oracle_y4m_parse_header   read_y4m_header (Source/App/EncApp/EbAppInputy4m.c:35-243)
 *   svt_oracle_pad_input_picture  pad_input_picture (Source/Lib/Common/Codec/EbMcp.c:273-317)
 *   svt_oracle_generate_padding   generate_padding / generate_padding16_bit (EbMcp.c:176-267)
 *   svt_oracle_decimation_2d      Decimation2D (EbPictureAnalysisProcess.c:170-195)
 *   svt_oracle_unpack8            un_pack8_bit_data (C_DEFAULT/EbPackUnPack_C.c:152-175)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "svt_oracle.h"

/* EbAppInputy4m.c:13-33: copy up to (not including) chr or a newline.  The reference copies with EB_STRNCPY(dst, src, count) =
 * strncpy_ss(dst, sizeof(dst), ...) where dst is a char POINTER: the bound is 8, and strncpy_ss (EbAppFifo.c:160-283) clears the
 * destination when the token does not fit with its terminator or is empty - so a token of 8 or more characters reads as "".
 * (That is why the reference's application rejects "C420mpeg2" and "C420paldv", which its own table lists.) */
static const char *copy_until(const char *src, char *dst, size_t cap, char chr) {
    size_t n = 0;
    const char *s0 = src;
    while (*src != chr && *src != '\n' && *src != '\0') { src++; n++; }
    if (n == 0 || n >= 8 || n + 1 > cap) dst[0] = '\0';
    else { memcpy(dst, s0, n); dst[n] = '\0'; }
    return src;
}

/* The header line after the "YUV4MPEG2" signature (EbAppInputy4m.c:35-243).  Returns 0, or -1 where the reference returns
 * EB_ErrorBadParameter (unknown interlace / chroma token, missing W / H / F). */
int svt_oracle_y4m_parse_header(const char *line, svt_oracle_y4m_info *out) {
    uint32_t bitdepth = 8, width = 0, height = 0, fr_n = 0, fr_d = 0;
    char chroma[8] = "420", scan = 'p', tok[128];
    int interlaced = 1;                                   /* the reference initialises interlaced = EB_TRUE (:43) */
    const char *p = line;
    for (; *p != '\0'; p++) {
        if (*p == 0x20) continue;
        switch (*p++) {
        case 'W': width = (uint32_t)strtol(p, (char **)&p, 10); break;
        case 'H': height = (uint32_t)strtol(p, (char **)&p, 10); break;
        case 'I':
            switch (*p++) {
            case 'p': interlaced = 0; scan = 'p'; break;
            case 't': interlaced = 1; scan = 't'; break;
            case 'b': interlaced = 1; scan = 'b'; break;
            default: return -1;
            }
            break;
        case 'C': {
            p = copy_until(p, tok, sizeof(tok), 0x20);
            static const struct { const char *name, *chroma; uint32_t bd; } fmt[] = {
                {"420mpeg2", "420", 8}, {"420paldv", "420", 8}, {"420jpeg", "420", 8},
                {"420p16", "420", 16}, {"422p16", "422", 16}, {"444p16", "444", 16},
                {"420p14", "420", 14}, {"422p14", "422", 14}, {"444p14", "444", 14},
                {"420p12", "420", 12}, {"422p12", "422", 12}, {"444p12", "444", 12},
                {"420p10", "420", 10}, {"422p10", "422", 10}, {"444p10", "444", 10},
                {"420p9", "420", 9}, {"422p9", "422", 9}, {"444p9", "444", 9},
                {"420", "420", 8}, {"411", "411", 8}, {"422", "422", 8}, {"444", "444", 8},
                {"mono16", "400", 16}, {"mono12", "400", 12}, {"mono10", "400", 10}, {"mono9", "400", 9}, {"mono", "400", 8}};
            size_t i, n = sizeof(fmt) / sizeof(fmt[0]);
            for (i = 0; i < n; i++)
                if (strcmp(fmt[i].name, tok) == 0) break;
            if (i == n) return -1;
            strcpy(chroma, fmt[i].chroma);
            bitdepth = fmt[i].bd;
        } break;
        case 'F':
            p = copy_until(p, tok, sizeof(tok), ':');
            fr_n = (uint32_t)strtol(tok, NULL, 10);
            p++;
            p = copy_until(p, tok, sizeof(tok), 0x20);
            fr_d = (uint32_t)strtol(tok, NULL, 10);
            break;
        case 'A':
            p = copy_until(p, tok, sizeof(tok), ':');
            p++;
            p = copy_until(p, tok, sizeof(tok), 0x20);
            break;
        default: break;
        }
        if (*p == '\0') break;                            /* (the tokens above may stop on the terminator) */
    }
    if (width == 0 || height == 0 || fr_n == 0 || fr_d == 0) return -1;
    memset(out, 0, sizeof(*out));
    out->width = width; out->height = height; out->fr_n = fr_n; out->fr_d = fr_d;
    out->bit_depth = bitdepth; out->interlaced = (uint32_t)interlaced; out->scan_type = scan;
    strcpy(out->chroma, chroma);
    return 0;
}

/* EbMcp.c:273-317.  es = bytes per sample (the reference is called per byte plane; a 16-bit plane pads whole samples) */
void svt_oracle_pad_input_picture(uint8_t *pic, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_right,
                                  uint32_t pad_bottom, int es) {
    for (uint32_t y = 0; y < h && pad_right; y++)
        for (uint32_t x = 0; x < pad_right; x++) memcpy(pic + ((size_t)y * stride + w + x) * es, pic + ((size_t)y * stride + w - 1) * es, es);
    for (uint32_t y = 0; y < pad_bottom; y++)
        memcpy(pic + (size_t)(h + y) * stride * es, pic + (size_t)(h - 1) * stride * es, (size_t)(w + pad_right) * es);
}

/* EbMcp.c:176-267: horizontal replication of every picture row, then whole rows (stride samples) copied up and down */
void svt_oracle_generate_padding(uint8_t *buf, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w, uint32_t pad_h, int es) {
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *row = buf + ((size_t)(pad_h + y) * stride + pad_w) * es;
        for (uint32_t x = 0; x < pad_w; x++) {
            memcpy(row - (size_t)(x + 1) * es, row, es);
            memcpy(row + (size_t)(w + x) * es, row + (size_t)(w - 1) * es, es);
        }
    }
    const uint8_t *top = buf + (size_t)pad_h * stride * es, *bot = buf + (size_t)(pad_h + h - 1) * stride * es;
    for (uint32_t y = 1; y <= pad_h; y++) {
        memcpy(buf + (size_t)(pad_h - y) * stride * es, top, (size_t)stride * es);
        memcpy(buf + (size_t)(pad_h + h - 1 + y) * stride * es, bot, (size_t)stride * es);
    }
}

/* EbPictureAnalysisProcess.c:170-195 */
void svt_oracle_decimation_2d(const uint8_t *in, uint32_t in_stride, uint32_t w, uint32_t h, uint8_t *out, uint32_t out_stride,
                              uint32_t step) {
    for (uint32_t y = 0; y < h; y += step) {
        for (uint32_t x = 0; x < w; x += step) out[x >> (step >> 1)] = in[x];
        in += (size_t)in_stride << (step >> 1);
        out += out_stride;
    }
}

/* C_DEFAULT/EbPackUnPack_C.c:152-175: the 8-bit plane of a 10-bit picture (the reference's shift is the literal 2) */
void svt_oracle_unpack8(const uint16_t *in, uint32_t in_stride, uint8_t *out, uint32_t out_stride, uint32_t w, uint32_t h) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) out[(size_t)y * out_stride + x] = (uint8_t)(in[(size_t)y * in_stride + x] >> 2);
}
