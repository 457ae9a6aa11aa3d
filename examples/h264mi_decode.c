/*
 * examples/h264mi_decode.c -- the C ABI from plain C: decode an Annex-B file to raw I420 on the GPU.
 *
 *   h264mi_decode in.h264 out.yuv [frames_per_batch]
 *
 * What a Go/cgo (or any FFI) caller does is exactly this sequence: cut the byte
 * stream at access units, hand whole access units to h264mi_decode_batch, read the frames back.
 * The access-unit cut below is the same rule as AccessUnitSplitter in h264decode_amd/h264.py
 * (a new picture starts at an access unit delimiter, at a parameter set / SEI that follows a
 * slice, or at the first slice of a new picture: h264mi_slice_starts_picture on the parsed slice
 * headers, or a slice that starts where a slice of the current picture already started -- with
 * slice groups or arbitrary slice order "first_mb_in_slice == 0" is not that test).
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

    h264mi_config cfg = H264MI_CONFIG_INIT; /* zero-initialised, struct_size = this build's sizeof */
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
    /* parameter sets by id and the slices of the current picture, for the picture-boundary test */
    static h264mi_sps sps_tab[32];
    static h264mi_pps pps_tab[256];
    static uint8_t sps_ok[32], pps_ok[256];
    static h264mi_slice_header first_hdr, hdr;
    static int32_t first_mbs[1024];
    int n_first = 0, have_first = 0;
    uint8_t *rb = (uint8_t *)malloc((size_t)len + 16);
    for (int i = 0; i <= n; i++) {
        int cut = i == n;
        if (!cut) {
            const int t = nals[i].type;
            const uint8_t *p = buf + nals[i].offset;
            size_t rl = 0;
            h264mi_nal nh;
            if (t == 7 || t == 8) {
                if (h264mi_nal_parse(p, (size_t)nals[i].num_bytes, &nh, rb, &rl) == H264MI_OK) {
                    if (t == 7) {
                        h264mi_sps tmp;
                        if (h264mi_sps_parse(rb, rl, &tmp) == H264MI_OK && tmp.id >= 0 && tmp.id < 32) sps_tab[tmp.id] = tmp, sps_ok[tmp.id] = 1;
                    } else
                        for (int k = 0; k < 32; k++) { /* seq_parameter_set_id is inside: let every SPS try */
                            h264mi_pps tmp;
                            if (sps_ok[k] && h264mi_pps_parse(&sps_tab[k], rb, rl, &tmp) == H264MI_OK && tmp.sps_id == k && tmp.id >= 0 && tmp.id < 256) {
                                pps_tab[tmp.id] = tmp, pps_ok[tmp.id] = 1;
                                break;
                            }
                        }
                }
            }
            if (t == 1 || t == 5) {
                int new_pic = -1;
                if (h264mi_nal_parse(p, (size_t)nals[i].num_bytes, &nh, rb, &rl) == H264MI_OK)
                    for (int k = 0; k < 256 && new_pic < 0; k++) /* pic_parameter_set_id is inside: let every PPS try */
                        if (pps_ok[k] && sps_ok[pps_tab[k].sps_id] &&
                            h264mi_slice_header_parse(&sps_tab[pps_tab[k].sps_id], &pps_tab[k], nh.ref_idc, nh.type, rb, rl, &hdr) == H264MI_OK && hdr.pps_id == k) {
                            new_pic = 0;
                            for (int m = 0; m < n_first; m++) new_pic |= first_mbs[m] == hdr.first_mb_in_slice;
                            if (have_first && h264mi_slice_starts_picture(&sps_tab[pps_tab[k].sps_id], &first_hdr, &hdr) == 1) new_pic = 1;
                        }
                if (new_pic < 0) { /* parameter sets unknown: first_mb_in_slice == 0 */
                    new_pic = (p[1] & 0x80) != 0;
                    memset(&hdr, 0, sizeof(hdr));
                    hdr.first_mb_in_slice = new_pic ? 0 : -1;
                }
                if (new_pic && seen_vcl) cut = 1;
                if (cut || !seen_vcl) n_first = 0, have_first = 0;
                if (!have_first) first_hdr = hdr, have_first = 1;
                if (n_first < 1024) first_mbs[n_first++] = hdr.first_mb_in_slice;
            } else if ((t >= 6 && t <= 9) && seen_vcl)
                cut = 1, n_first = 0, have_first = 0;
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
