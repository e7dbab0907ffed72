/*
 * oracle/ref_picture.c — OUR harness around the reference's own picture-input functions; compiled into
 * oracle/_ref/libsvtref.so (oracle/Makefile).  TEST INFRASTRUCTURE ONLY.  Nothing of the reference is re-implemented:
 *   ref_y4m_header         check_if_y4m + read_y4m_header (Source/App/EncApp/EbAppInputy4m.c:269-290, 35-243) on a file, through
 *                          the application's own EbConfig; ref_y4m_frame_delimiter = read_y4m_frame_delimiter (:247-266)
 *   ref_generate_padding   generate_padding / generate_padding16_bit (Source/Lib/Common/Codec/EbMcp.c:176-267)
 *   ref_pad_input_picture  pad_input_picture (EbMcp.c:273-317)
 *   ref_decimation_2d      Decimation2D (EbPictureAnalysisProcess.c:170-195)
 *   ref_unpack8            un_pack8_bit_data (C_DEFAULT/EbPackUnPack_C.c:152-175)
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "EbAppConfig.h"
#include "EbAppInputy4m.h"

void generate_padding(uint8_t *src_pic, uint32_t src_stride, uint32_t original_src_width, uint32_t original_src_height,
                      uint32_t padding_width, uint32_t padding_height);
void generate_padding16_bit(uint8_t *src_pic, uint32_t src_stride, uint32_t original_src_width, uint32_t original_src_height,
                            uint32_t padding_width, uint32_t padding_height);
void pad_input_picture(uint8_t *src_pic, uint32_t src_stride, uint32_t original_src_width, uint32_t original_src_height,
                       uint32_t pad_right, uint32_t pad_bottom);
void Decimation2D(uint8_t *input_samples, uint32_t input_stride, uint32_t input_area_width, uint32_t input_area_height,
                  uint8_t *decimSamples, uint32_t decimStride, uint32_t decimStep);

void un_pack8_bit_data(uint16_t *in16_bit_buffer, uint32_t in_stride, uint8_t *out8_bit_buffer, uint32_t out8_stride, uint32_t width,
                       uint32_t height);

/* out[0..7] = is_y4m, rc, width, height, fr_n, fr_d, bit depth, interlaced; returns the file offset after the header */
long ref_y4m_header(const char *path, int32_t *out) {
    EbConfig *cfg = (EbConfig *)calloc(1, sizeof(EbConfig));
    long pos = -1;
    memset(out, 0, 8 * sizeof(int32_t));
    cfg->input_file = fopen(path, "rb");
    cfg->error_log_file = stderr;
    if (cfg->input_file) {
        out[0] = check_if_y4m(cfg) == EB_TRUE;
        if (out[0]) {
            out[1] = read_y4m_header(cfg);
            out[2] = (int32_t)cfg->source_width; out[3] = (int32_t)cfg->source_height;
            out[4] = (int32_t)cfg->frame_rate_numerator; out[5] = (int32_t)cfg->frame_rate_denominator;
            out[6] = (int32_t)cfg->encoder_bit_depth; out[7] = (int32_t)cfg->interlaced_video;
            if (out[1] == 0) {
                out[1] = read_y4m_frame_delimiter(cfg);         /* the first "FRAME\n" */
                pos = ftell(cfg->input_file);
            }
        }
        fclose(cfg->input_file);
    }
    free(cfg);
    return pos;
}

void ref_generate_padding(uint8_t *buf, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_w, uint32_t pad_h, int is16) {
    if (is16) generate_padding16_bit(buf, stride << 1, w << 1, h, pad_w << 1, pad_h);   /* byte units, as its callers pass them */
    else generate_padding(buf, stride, w, h, pad_w, pad_h);
}
void ref_pad_input_picture(uint8_t *pic, uint32_t stride, uint32_t w, uint32_t h, uint32_t pad_right, uint32_t pad_bottom) {
    pad_input_picture(pic, stride, w, h, pad_right, pad_bottom);
}
void ref_decimation_2d(uint8_t *in, uint32_t in_stride, uint32_t w, uint32_t h, uint8_t *out, uint32_t out_stride, uint32_t step) {
    Decimation2D(in, in_stride, w, h, out, out_stride, step);
}
void ref_unpack8(uint16_t *in, uint32_t in_stride, uint8_t *out, uint32_t out_stride, uint32_t w, uint32_t h) {
    un_pack8_bit_data(in, in_stride, out, out_stride, w, h);
}
