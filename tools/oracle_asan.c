#include "h264o.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static unsigned long long s = 88172645463325252ull;
static unsigned rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (unsigned)(s >> 11); }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(len), *m = malloc(len); if (fread(buf, 1, len, f) != (size_t)len) return 1; fclose(f);
    int iters = argc > 2 ? atoi(argv[2]) : 100; s ^= argc > 3 ? atoi(argv[3]) * 7919ull : 0;
    size_t cap = 64u << 20; uint8_t *out = malloc(cap);
    int okc = 0, bad = 0;
    for (int it = 0; it < iters; it++) {
        memcpy(m, buf, len); long L = len;
        int flips = it == 0 ? 0 : 1 + rnd() % 12;
        for (int k = 0; k < flips; k++) m[rnd() % len] ^= 1u << (rnd() % 8);
        if (it && rnd() % 4 == 0) L = 1 + rnd() % len;
        h264o_decoder *d = h264o_decoder_create();
        h264o_stream_info info;
        int r = h264o_decode_stream(d, m, L, rnd() & 1, out, cap, &info);
        if (r == 0) okc++; else bad++;
        h264o_decoder_destroy(d);
    }
    printf("%s: %d decoded, %d rejected\n", argv[1], okc, bad);
    return 0;
}
