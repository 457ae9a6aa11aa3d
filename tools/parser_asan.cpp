// CPU-only ASan/UBSan harness for the host parsers (mi_parse.cpp has no HIP dependency)
#include "mi_parse.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
using namespace mi;
// MSB-first bit writer for hand-made parameter sets
struct BW {
    std::vector<uint8_t> v; int nbits = 0;
    void u(int n, uint64_t x) { for (int i = n - 1; i >= 0; i--) { if (nbits % 8 == 0) v.push_back(0); if ((x >> i) & 1) v.back() |= 0x80 >> (nbits % 8); nbits++; } }
    void ue(uint64_t x) { int n = 0; while (((x + 1) >> (n + 1)) != 0) n++; u(n, 0); u(n + 1, x + 1); }
    void se(int64_t x) { ue(x > 0 ? 2 * x - 1 : -2 * x); }
    void trail() { u(1, 1); while (nbits % 8) u(1, 0); }
};
// insert `code` (nb bits) at bit position `at` of `src`
static std::vector<uint8_t> splice_bits(const std::vector<uint8_t> &src, size_t at, uint64_t hi, uint64_t lo /* 31 zeros + '1' | 31 suffix bits */) {
    BW w;
    const size_t total = src.size() * 8;
    for (size_t i = 0; i < at && i < total; i++) w.u(1, (src[i / 8] >> (7 - i % 8)) & 1);
    w.u(32, hi), w.u(31, lo);
    for (size_t i = at; i < total; i++) w.u(1, (src[i / 8] >> (7 - i % 8)) & 1);
    return w.v;
}
// Round-3 advisor reproducers: ue(v) values >= 2^31 in the slice-group syntax of a PPS / in the ids.  Each must be REJECTED by the parser,
// and the map builder -- a public entry point of its own -- must refuse a hand-filled PPS with the same values instead of walking off the map.
static int reproducers(const h264mi_sps &sps) {
    int rejected = 0;
    std::vector<uint8_t> map(1 << 20), ids(1 << 20);
    size_t q = 0, nn = 0;
    h264mi_pps t;
    { // (1) map type 2, top_left = 0xFFFFFFFE / 0x80000005
        for (uint64_t tl : {0xFFFFFFFEull, 0x80000005ull}) {
            BW w; w.ue(0); w.ue(0); w.u(1, 0); w.u(1, 0); w.ue(1); w.ue(2); w.ue(tl); w.ue(tl + 1 > 0xFFFFFFFEull ? tl : tl + 1);
            w.ue(0); w.ue(0); w.u(1, 0); w.u(2, 0); w.se(0); w.se(0); w.se(0); w.u(3, 0); w.trail();
            int r = parse_pps_ids(&sps, w.v.data(), w.v.size(), &t, ids.data(), ids.size(), &q);
            if (r == 0) mb_to_slice_group_map(&sps, &t, ids.data(), q, 0, 0, map.data(), map.size(), &nn); else rejected++;
            h264mi_pps h; memset(&h, 0, sizeof(h)); h.num_slice_groups_minus1 = 1, h.slice_group_map_type = 2, h.top_left[0] = (int32_t)tl, h.bottom_right[0] = (int32_t)tl;
            if (map_unit_to_slice_group_map(&sps, &h, nullptr, 0, 0, map.data(), map.size(), &nn) != 0) rejected++;
        }
    }
    { // (2) map type 0, run_length_minus1 = {0, 0xFFFFFFFE}: the interleave loop stepped backwards for ever
        BW w; w.ue(0); w.ue(0); w.u(1, 0); w.u(1, 0); w.ue(1); w.ue(0); w.ue(0); w.ue(0xFFFFFFFEull);
        w.ue(0); w.ue(0); w.u(1, 0); w.u(2, 0); w.se(0); w.se(0); w.se(0); w.u(3, 0); w.trail();
        int r = parse_pps_ids(&sps, w.v.data(), w.v.size(), &t, ids.data(), ids.size(), &q);
        if (r == 0) mb_to_slice_group_map(&sps, &t, ids.data(), q, 0, 0, map.data(), map.size(), &nn); else rejected++;
        h264mi_pps h; memset(&h, 0, sizeof(h)); h.num_slice_groups_minus1 = 1, h.slice_group_map_type = 0, h.run_length_minus1[1] = (int32_t)0xFFFFFFFEu;
        if (map_unit_to_slice_group_map(&sps, &h, nullptr, 0, 0, map.data(), map.size(), &nn) != 0) rejected++;
    }
    { // (3) pic_parameter_set_id = 0x80000000: negative as int32, used to pass "id > 255" and index the per-stream PPS table
        BW w; w.ue(0x80000000ull); w.ue(0); w.u(1, 0); w.u(1, 0); w.ue(0); w.ue(0); w.ue(0); w.u(1, 0); w.u(2, 0); w.se(0); w.se(0); w.se(0); w.u(3, 0); w.trail();
        int r = parse_pps_ids(&sps, w.v.data(), w.v.size(), &t, ids.data(), ids.size(), &q);
        if (r != 0 || (t.id >= 0 && t.id <= 255)) rejected += r != 0;
    }
    return rejected;
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); long len = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf(len); if (fread(buf.data(), 1, len, f) != (size_t)len) return 1; fclose(f);
    int iters = argc > 2 ? atoi(argv[2]) : 20000;
    std::mt19937 rng(argc > 3 ? atoi(argv[3]) : 1);
    std::vector<h264mi_nal> nals(4096);
    int n = 0;
    annexb_scan(buf.data(), buf.size(), nals.data(), (int)nals.size(), &n);
    // collect SPS, PPS, first slices
    std::vector<std::vector<uint8_t>> sps_r, pps_r, sl_r; std::vector<int> sl_ref, sl_type;
    for (int i = 0; i < n; i++) {
        std::vector<uint8_t> r(nals[i].num_bytes + 8); size_t rl = 0; h264mi_nal h;
        if (nal_parse(buf.data() + nals[i].offset, nals[i].num_bytes, &h, r.data(), &rl) != 0) continue;
        r.resize(rl);
        if (h.type == 7) sps_r.push_back(r); else if (h.type == 8) pps_r.push_back(r); else if (h.type == 1 || h.type == 5) { sl_r.push_back(r); sl_ref.push_back(h.ref_idc); sl_type.push_back(h.type); }
    }
    h264mi_sps sps; h264mi_pps pps; h264mi_slice_header sh;
    if (sps_r.empty() || parse_sps(sps_r[0].data(), sps_r[0].size(), &sps) != 0) { printf("no sps\n"); return 0; }
    std::vector<uint8_t> ids(1 << 20); size_t nids = 0;
    if (pps_r.empty() || parse_pps_ids(&sps, pps_r[0].data(), pps_r[0].size(), &pps, ids.data(), ids.size(), &nids) != 0) { printf("no pps\n"); return 0; }
    long ok = 0, bad = 0;
    std::vector<uint8_t> map(1 << 20);
    const int rep = reproducers(sps);
    printf("reproducers: %d of 7 refused\n", rep);
    if (rep != 7) return 2;
    // maximal Exp-Golomb codes (31 leading zeros: values 2^31 - 1 .. 2^32 - 2) spliced in at EVERY bit position of every parameter set and of the
    // first 160 bits of every slice header: whatever field starts there reads a value that is negative as int32
    long spliced = 0;
    for (int which = 0; which < 3; which++) {
        const auto &set = which == 0 ? sps_r : which == 1 ? pps_r : sl_r;
        for (size_t k = 0; k < set.size() && k < 6; k++)
            for (size_t at = 0; at < std::min<size_t>(set[k].size() * 8, which == 2 ? 160 : 400); at++)
                for (uint64_t lo : {0x7FFFFFFFull, 0ull, 1ull, 6ull}) {
                    std::vector<uint8_t> m = splice_bits(set[k], at, 1, lo);
                    int r;
                    if (which == 0) { h264mi_sps t; r = parse_sps(m.data(), m.size(), &t); if (r == 0) { h264mi_pps tp; size_t q; parse_pps_ids(&t, pps_r[0].data(), pps_r[0].size(), &tp, ids.data(), ids.size(), &q); } }
                    else if (which == 1) { h264mi_pps t; size_t q = 0, nn; r = parse_pps_ids(&sps, m.data(), m.size(), &t, ids.data(), ids.size(), &q);
                        if (r == 0) { if (t.id < 0 || t.id > 255 || t.sps_id < 0 || t.sps_id > 31) return 3; mb_to_slice_group_map(&sps, &t, ids.data(), q, 3, 0, map.data(), map.size(), &nn); mb_to_slice_group_map(&sps, &t, ids.data(), q, 3, 1, map.data(), map.size(), &nn); } }
                    else { r = parse_slice_header(&sps, &pps, sl_ref[k], sl_type[k], m.data(), m.size(), &sh);
                        if (r == 0 && (sh.first_mb_in_slice < 0 || sh.pps_id < 0 || sh.num_ref_idx_l0_active_minus1 < 0 || sh.num_ref_idx_l0_active_minus1 > 31 || sh.luma_log2_weight_denom < 0 || sh.luma_log2_weight_denom > 7 || sh.cabac_init < 0 || sh.disable_deblocking_filter < 0)) return 4; }
                    spliced++;
                }
    }
    printf("spliced maximal ue(v) codes: %ld inputs\n", spliced);
    for (int it = 0; it < iters; it++) {
        int which = rng() % 4;
        const std::vector<uint8_t> &src = which == 0 ? sps_r[rng() % sps_r.size()] : which == 1 ? pps_r[rng() % pps_r.size()] : sl_r[rng() % sl_r.size()];
        std::vector<uint8_t> m(src.begin(), src.begin() + (which >= 2 ? std::min<size_t>(src.size(), 96) : src.size()));
        int flips = 1 + rng() % 6;
        for (int k = 0; k < flips; k++) m[rng() % m.size()] ^= 1u << (rng() % 8);
        if (rng() % 5 == 0) m.resize(1 + rng() % m.size());
        int r;
        if (which == 0) { h264mi_sps t; r = parse_sps(m.data(), m.size(), &t); if (r == 0 && rng() % 3 == 0) { h264mi_pps tp; size_t q; parse_pps_ids(&t, pps_r[0].data(), pps_r[0].size(), &tp, ids.data(), ids.size(), &q); } }
        else if (which == 1) { h264mi_pps t; size_t q = 0; r = parse_pps_ids(&sps, m.data(), m.size(), &t, ids.data(), ids.size(), &q);
            if (r == 0 && t.num_slice_groups_minus1 > 0) { size_t nn; mb_to_slice_group_map(&sps, &t, ids.data(), q, rng() % 40, 0, map.data(), map.size(), &nn); } }
        else { int i = rng() % sl_r.size(); r = parse_slice_header(&sps, &pps, sl_ref[i], sl_type[i], m.data(), m.size(), &sh);
            if (r == 0 && pps.num_slice_groups_minus1 > 0) { size_t nn; mb_to_slice_group_map(&sps, &pps, ids.data(), nids, sh.slice_group_change_cycle, 0, map.data(), map.size(), &nn); next_mb_address(map.data(), nn, rng() % (nn + 2)); } }
        (r == 0 ? ok : bad)++;
        // annexb_scan + nal_parse on mutated whole-stream windows
        if (it % 50 == 0) {
            size_t o = rng() % buf.size(), l = std::min<size_t>(buf.size() - o, 1 + rng() % 4000);
            std::vector<uint8_t> w(buf.begin() + o, buf.begin() + o + l);
            for (int k = 0; k < 8; k++) w[rng() % w.size()] ^= 1u << (rng() % 8);
            int nn = 0; annexb_scan(w.data(), w.size(), nals.data(), (int)nals.size(), &nn);
            for (int i = 0; i < nn; i++) { std::vector<uint8_t> r(nals[i].num_bytes + 8); size_t rl; h264mi_nal h; nal_parse(w.data() + nals[i].offset, nals[i].num_bytes, &h, r.data(), &rl); }
        }
    }
    printf("parser fuzz: %ld accepted, %ld rejected\n", ok, bad);
    return 0;
}
