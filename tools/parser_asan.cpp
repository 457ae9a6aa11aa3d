// CPU-only ASan/UBSan harness for the host parsers (mi_parse.cpp has no HIP dependency)
#include "mi_parse.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
using namespace mi;
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
