/*
 * oracle/h264o_bits.c -- bit reader, Exp-Golomb (9.1), Annex-B scan (B.1), NAL/RBSP (7.3.1).
 * TEST INFRASTRUCTURE ONLY (see h264o.h).
 */
#include <string.h>
#include "h264o.h"

void h264o_br_init(h264o_br *b, const uint8_t *p, size_t nbytes) {
    b->p = p;
    b->nbits = (int64_t)nbytes * 8;
    b->pos = 0;
    b->err = 0;
}

static inline uint32_t br_bit(h264o_br *b) {
    if (b->pos >= b->nbits) {
        b->err = 1;
        b->pos++;
        return 0;
    }
    uint32_t v = (b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1;
    b->pos++;
    return v;
}

/* u(n): MSB first.  Follows h264/bit_reader.go:292-325 (Read/NextField) minus the per-bit
 * allocations and the off-by-one bounds check (Appendix A30). */
uint32_t h264o_u(h264o_br *b, int n) {
    uint32_t v = 0;
    for (int i = 0; i < n; i++) v = (v << 1) | br_bit(b);
    return v;
}

uint32_t h264o_peek(h264o_br *b, int n) {
    uint32_t v = 0;
    int64_t pos = b->pos;
    for (int i = 0; i < n; i++, pos++) {
        uint32_t bit = 0;
        if (pos < b->nbits) bit = (b->p[pos >> 3] >> (7 - (pos & 7))) & 1;
        v = (v << 1) | bit;
    }
    return v;
}

void h264o_skip(h264o_br *b, int n) {
    b->pos += n;
    if (b->pos > b->nbits) b->err = 1;
}

/* ue(v), 9.1: leadingZeroBits, then codeNum = 2^lz - 1 + read_bits(lz).
 * h264/bit_reader.go:62-64,174-196. */
uint32_t h264o_ue(h264o_br *b) {
    int lz = 0;
    while (br_bit(b) == 0) {
        lz++;
        if (lz > 32 || b->err) {
            b->err = 1;
            return 0;
        }
    }
    if (lz == 0) return 0;
    if (lz == 32) return 0xFFFFFFFFu; /* codeNum 2^32-1: only legal as an escape, never in scope */
    return ((1u << lz) - 1) + h264o_u(b, lz);
}

/* se(v), 9.1.1: (-1)^(k+1) * Ceil(k/2).  The reference's version (h264/bit_reader.go:158-161)
 * is one too small for every odd k (Appendix A3); not reproduced. */
int32_t h264o_se(h264o_br *b) {
    uint32_t k = h264o_ue(b);
    int32_t v = (int32_t)((k + 1) >> 1);
    return (k & 1) ? v : -v;
}

/* te(v), 9.1: range==1 -> inverted single bit.  (h264/bit_reader.go:147-155 never reaches the
 * 1-bit case: Appendix A28.) */
uint32_t h264o_te(h264o_br *b, int range) {
    if (range > 1) return h264o_ue(b);
    return !br_bit(b);
}

/* more_rbsp_data(), 7.2: true when something other than rbsp_trailing_bits follows.
 * Non-destructive (the reference's h264/bit_reader.go:199-219 consumes bits: Appendix A29). */
int h264o_more_rbsp_data(h264o_br *b) {
    if (b->pos >= b->nbits) return 0;
    /* find last 1 bit in the buffer */
    int64_t last = b->nbits - 1;
    while (last >= 0 && ((b->p[last >> 3] >> (7 - (last & 7))) & 1) == 0) last--;
    if (last < 0) return 0;
    return b->pos < last;
}

/* Annex B.1 byte stream scan: NALs are delimited by 00 00 01 (optionally preceded by zero
 * bytes); trailing_zero_8bits are stripped.  The reference only recognises 00 00 00 01 and
 * keeps the next start code inside each NAL (h264/server.go:28-39,64-111; Appendix A31). */
int h264o_annexb_scan(const uint8_t *buf, size_t len, h264o_nal *out, int cap) {
    int n = 0;
    size_t i = 0;
    size_t start = (size_t)-1;
    while (i + 2 < len) {
        if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] == 1) {
            if (start != (size_t)-1) {
                size_t end = i;
                while (end > start && buf[end - 1] == 0) end--;
                if (n < cap && end > start) {
                    out[n].offset = start;
                    out[n].size = end - start;
                    out[n].forbidden_zero_bit = buf[start] >> 7;
                    out[n].nal_ref_idc = (buf[start] >> 5) & 3;
                    out[n].nal_unit_type = buf[start] & 31;
                    n++;
                }
            }
            start = i + 3;
            i += 3;
        } else {
            i++;
        }
    }
    if (start != (size_t)-1 && start < len) {
        size_t end = len;
        while (end > start && buf[end - 1] == 0) end--;
        if (n < cap && end > start) {
            out[n].offset = start;
            out[n].size = end - start;
            out[n].forbidden_zero_bit = buf[start] >> 7;
            out[n].nal_ref_idc = (buf[start] >> 5) & 3;
            out[n].nal_unit_type = buf[start] & 31;
            n++;
        }
    }
    return n;
}

/* 7.3.1 / 7.4.1.1: drop each emulation_prevention_three_byte (00 00 03 -> 00 00).
 * h264/nalUnit.go:32-37,106-126 (which stops 3 bytes early: Appendix A31). */
size_t h264o_nal_to_rbsp(const uint8_t *nal, size_t size, uint8_t *rbsp) {
    size_t o = 0;
    int zeros = 0;
    for (size_t i = 1; i < size; i++) {
        uint8_t c = nal[i];
        if (zeros >= 2 && c == 3) {
            zeros = 0;
            continue;
        }
        rbsp[o++] = c;
        if (c == 0)
            zeros++;
        else
            zeros = 0;
    }
    return o;
}

int h264o_kat_ue(const uint8_t *bytes, size_t n, int count, uint32_t *out) {
    h264o_br b;
    h264o_br_init(&b, bytes, n);
    for (int i = 0; i < count; i++) out[i] = h264o_ue(&b);
    return b.err ? -1 : (int)b.pos;
}
int h264o_kat_se(const uint8_t *bytes, size_t n, int count, int32_t *out) {
    h264o_br b;
    h264o_br_init(&b, bytes, n);
    for (int i = 0; i < count; i++) out[i] = h264o_se(&b);
    return b.err ? -1 : (int)b.pos;
}
