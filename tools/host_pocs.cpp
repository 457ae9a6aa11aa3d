// tools/host_pocs.cpp -- TEST TOOLING: the host side of libh264mi built against tools/hoststub (a null device: kernels are not
// run) decodes one Annex-B file in a single batch and prints, per frame in decoding order, what its picture management made of
// it: "pic_order_cnt frame_num nal_ref_idc idr new_sequence", then the output order of the batch.  tests/test_host_picture_management.py holds these against the
// generator's intent on the CPU -- the product's own 8.2.1 code without a GPU.
#include "h264mi.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: host_pocs stream.h264 max_width max_height max_frames [max_slices]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(len);
    if (fread(buf.data(), 1, len, f) != static_cast<size_t>(len)) return 2;
    fclose(f);
    h264mi_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.struct_size = sizeof cfg;
    cfg.max_streams = 1, cfg.max_width = atoi(argv[2]), cfg.max_height = atoi(argv[3]), cfg.max_frames_per_batch = atoi(argv[4]);
    cfg.max_slices_per_frame = argc > 5 ? atoi(argv[5]) : 64, cfg.max_bitstream_bytes = len + 4096;
    h264mi_decoder *dec = nullptr;
    if (h264mi_decoder_create(&cfg, &dec) != 0) { fprintf(stderr, "create: %s\n", h264mi_last_error_string()); return 1; }
    const uint8_t *bufs[1] = {buf.data()};
    size_t lens[1] = {buf.size()};
    h264mi_batch_info info;
    int32_t r = h264mi_batch_prepare(dec, 1, bufs, lens, &info), st = 0;
    h264mi_stream_status(dec, 0, &st);
    if (r != 0 || st != 0) { fprintf(stderr, "prepare: %d, stream status %d: %s\n", r, st, h264mi_last_error_string()); return 1; }
    if (h264mi_batch_execute(dec) != 0 || h264mi_batch_sync(dec) != 0) { fprintf(stderr, "execute: %s\n", h264mi_last_error_string()); return 1; }
    int32_t n = 0;
    h264mi_stream_frame_count(dec, 0, &n);
    for (int i = 0; i < n; i++) {
        h264mi_frame_info fi;
        if (h264mi_frame_get_info(dec, 0, i, &fi) != 0) return 1;
        printf("%d %d %d %d %d\n", fi.pic_order_cnt, fi.frame_num, fi.nal_ref_idc, fi.idr, fi.new_sequence);
    }
    // the output order of the batch (h264mi_stream_output_order): one line "order i j k ..."
    std::vector<int32_t> order(n > 0 ? n : 1);
    int32_t no = 0;
    if (h264mi_stream_output_order(dec, 0, order.data(), n, &no) != 0 || no != n) return 1;
    printf("order");
    for (int i = 0; i < no; i++) printf(" %d", order[i]);
    printf("\n");
    h264mi_decoder_destroy(dec);
    return 0;
}
