/*
 * examples/h264mi_decode.c -- the C ABI from plain C: decode an Annex-B file to raw I420 on the GPU.
 *
 *   h264mi_decode in.h264 out.yuv [frames_per_batch]
 *
 * What a Go/cgo (or any FFI) caller does is exactly this sequence: cut the byte
 * stream at access units, hand whole access units to h264mi_decode_batch, read the frames back.
 * The access-unit cut below is the same rule as AccessUnitSplitter in h264decode_amd/h264.py
 * (a new picture starts at an access unit delimiter, at a parameter set / SEI that follows a
 * slice, or at a slice whose first_mb_in_slice is 0).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "h264mi.h"

static int fail(const char *what, int code) {
    fprintf(stderr, "%s failed: %d (%s)\n", what, code, h264mi_last_error_string());
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s in.h264 out.yuv [frames_per_batch]\n", argv[0]);
        return 2;
    }
    const int per_batch = argc > 3 ? atoi(argv[3]) : 30;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return fail("fopen", -1);
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *buf = (uint8_t *)malloc((size_t)len);
    if (fread(buf, 1, (size_t)len, f) != (size_t)len) return fail("fread", -1);
    fclose(f);

    /* NAL table + picture size from the first SPS */
    int32_t cap = 1 << 16, n = 0, r;
    h264mi_nal *nals = (h264mi_nal *)malloc(sizeof(h264mi_nal) * (size_t)cap);
    while ((r = h264mi_annexb_scan(buf, (size_t)len, nals, cap, &n)) == H264MI_ECAPACITY) {
        cap *= 4;
        nals = (h264mi_nal *)realloc(nals, sizeof(h264mi_nal) * (size_t)cap);
    }
    if (r != H264MI_OK) return fail("h264mi_annexb_scan", r);
    h264mi_sps sps;
    memset(&sps, 0, sizeof(sps));
    for (int i = 0; i < n; i++)
        if (nals[i].type == 7) {
            uint8_t *rbsp = (uint8_t *)malloc((size_t)nals[i].num_bytes);
            size_t rl = 0;
            h264mi_nal hdr;
            if ((r = h264mi_nal_parse(buf + nals[i].offset, (size_t)nals[i].num_bytes, &hdr, rbsp, &rl)) != H264MI_OK) return fail("h264mi_nal_parse", r);
            if ((r = h264mi_sps_parse(rbsp, rl, &sps)) != H264MI_OK) return fail("h264mi_sps_parse", r);
            free(rbsp);
            break;
        }
    if (!sps.width) return fail("no SPS found", H264MI_EBITSTREAM);

    h264mi_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.max_streams = 1, cfg.max_width = sps.width, cfg.max_height = sps.height;
    cfg.max_frames_per_batch = per_batch, cfg.max_slices_per_frame = 32, cfg.max_bitstream_bytes = len + (1 << 20);
    h264mi_decoder *dec = NULL;
    if ((r = h264mi_decoder_create(&cfg, &dec)) != H264MI_OK) return fail("h264mi_decoder_create", r);

    FILE *out = fopen(argv[2], "wb");
    const size_t fsz = (size_t)sps.width * sps.height * 3 / 2;
    uint8_t *frame = NULL;
    size_t frame_cap = 0;
    long total = 0;
    int start = 0, pics = 0, seen_vcl = 0;
    for (int i = 0; i <= n; i++) {
        int cut = i == n;
        if (!cut) {
            const int t = nals[i].type;
            const uint8_t *p = buf + nals[i].offset;
            if (t == 1 || t == 5) {
                if ((p[1] & 0x80) && seen_vcl) cut = 1; /* first_mb_in_slice == 0: next picture */
            } else if ((t >= 6 && t <= 9) && seen_vcl)
                cut = 1;
        }
        if (cut && seen_vcl) {
            pics++;
            seen_vcl = 0;
            if (pics == per_batch || i == n) { /* decode nals[start .. i) */
                const size_t b0 = (size_t)nals[start].offset >= 4 ? (size_t)nals[start].offset - 4 : 0; /* include the start code */
                const size_t b1 = i == n ? (size_t)len : (size_t)nals[i].offset - 3;
                const uint8_t *ptr = buf + b0;
                const size_t l = b1 - b0;
                h264mi_batch_info info;
                if ((r = h264mi_decode_batch(dec, 1, &ptr, &l, &info)) != H264MI_OK) return fail("h264mi_decode_batch", r);
                int32_t k = 0;
                h264mi_stream_frame_count(dec, 0, &k);
                const size_t need = (size_t)info.coded_width * info.coded_height * 3 / 2;
                if (need > frame_cap) frame = (uint8_t *)realloc(frame, frame_cap = need);
                for (int fidx = 0; fidx < k; fidx++) {
                    if ((r = h264mi_frame_read(dec, 0, fidx, 1, frame, frame_cap)) != H264MI_OK) return fail("h264mi_frame_read", r);
                    fwrite(frame, 1, fsz, out);
                }
                total += k;
                start = i, pics = 0;
            }
        }
        if (i < n && (nals[i].type == 1 || nals[i].type == 5)) seen_vcl = 1;
    }
    fclose(out);
    h264mi_decoder_destroy(dec);
    fprintf(stderr, "%s: %ld frames %dx%d -> %s\n", h264mi_version(), total, sps.width, sps.height, argv[2]);
    return 0;
}
