// tools/host_asan.cpp -- TEST TOOLING: drives the host side of libh264mi (built against tools/hoststub: a null device,
// kernels not run) under AddressSanitizer / UBSan with intact and damaged streams, in the call orders a client uses:
// prepare / execute / sync in chunks of whole access units, several streams side by side, per-stream resets, isolation
// on and off, frame queries and read-back.  Built and run by tools/host_asan.sh.
#include "h264mi.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

static std::vector<uint8_t> slurp(const char *p) {
    std::vector<uint8_t> v;
    FILE *f = fopen(p, "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    v.resize(n);
    if (fread(v.data(), 1, n, f) != static_cast<size_t>(n)) v.clear();
    fclose(f);
    return v;
}

// cut positions in front of the NAL units that may begin an access unit (SPS, or a slice with first_mb_in_slice == 0)
static std::vector<size_t> au_cuts(const std::vector<uint8_t> &s) {
    std::vector<h264mi_nal> nals(8192);
    int32_t n = 0;
    h264mi_annexb_scan(s.data(), s.size(), nals.data(), static_cast<int32_t>(nals.size()), &n);
    std::vector<size_t> cuts;
    bool in_ps = false;
    for (int i = 0; i < n; i++) {
        const uint8_t *p = s.data() + nals[i].offset;
        const int type = p[0] & 31;
        const bool ps = type == 7 || type == 8 || type == 6 || type == 9;
        const bool first_slice = (type == 1 || type == 5) && nals[i].num_bytes > 1 && (p[1] & 0x80);
        if ((ps && !in_ps) || (first_slice && !in_ps)) {
            size_t o = nals[i].offset;
            while (o > 0 && s[o - 1] == 0) o--;  // back over the start code
            if (o >= 1 && s[o] == 0) cuts.push_back(o);
            else if (nals[i].offset >= 3) cuts.push_back(nals[i].offset - 3);
        }
        in_ps = ps;
    }
    if (cuts.empty() || cuts[0] != 0) cuts.insert(cuts.begin(), 0);
    cuts.push_back(s.size());
    return cuts;
}

static std::vector<std::vector<uint8_t>> streams;
static std::atomic<long> batches{0}, ok_streams{0}, failed_streams{0}, frames{0}, refused{0};
static std::atomic<long> codes[64];

// one client: `iters` decoders one after the other, each fed its own random mix of streams
static void client(int iters, unsigned seed) {
    std::mt19937 rng(seed);
    std::vector<uint8_t> pix(4 << 20);
    for (int it = 0; it < iters; it++) {
        const int ns = 1 + rng() % 4;
        h264mi_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = rng() % 8 == 0 ? offsetof(h264mi_config, b_pictures) : sizeof cfg; // (now and then a caller built against this header before `b_pictures` was added)
        if (rng() % 5 == 0) cfg.b_pictures = 1;
        cfg.max_streams = ns;
        const bool tight = rng() % 3 == 0;  // a decoder sized too small for some of what it will be given
        cfg.max_width = tight ? 64 + 16 * (rng() % 12) : 352;
        cfg.max_height = tight ? 64 + 16 * (rng() % 10) : 288;
        cfg.max_frames_per_batch = 1 + rng() % 8;
        cfg.max_slices_per_frame = tight ? 1 + rng() % 40 : 64;
        cfg.max_bitstream_bytes = tight ? 4096 << (rng() % 9) : 4 << 20;
        if (rng() % 4 == 0) cfg.max_ref_frames = 1 + rng() % 16;
        if (rng() % 4 == 0) cfg.coef_blocks_per_mb = 1 + rng() % 26;
        h264mi_decoder *dec = nullptr;
        if (h264mi_decoder_create(&cfg, &dec) != 0 || !dec) { refused++; continue; }
        if (rng() % 3 == 0) h264mi_decoder_set_isolation(dec, rng() & 1);
        // per stream: a (possibly damaged) copy and its cut list
        std::vector<std::vector<uint8_t>> data(ns);
        std::vector<std::vector<size_t>> cuts(ns);
        std::vector<size_t> pos(ns, 0);
        const bool damage = it % 3 != 0;
        for (int s = 0; s < ns; s++) {
            data[s] = streams[rng() % streams.size()];
            cuts[s] = au_cuts(data[s]);
            if (damage && rng() % 4 != 0) {
                const int flips = 1 + rng() % 10;
                for (int k = 0; k < flips; k++) data[s][rng() % data[s].size()] ^= 1u << (rng() % 8);
                if (rng() % 5 == 0) {  // splice a window of another stream in
                    const auto &o = streams[rng() % streams.size()];
                    size_t l = 1 + rng() % std::min<size_t>(o.size(), 600), so = rng() % (o.size() - l + 1), d = rng() % data[s].size();
                    l = std::min(l, data[s].size() - d);
                    memcpy(data[s].data() + d, o.data() + so, l);
                }
            }
        }
        for (int round = 0; round < 40; round++) {
            std::vector<const uint8_t *> bufs(ns, nullptr);
            std::vector<size_t> lens(ns, 0);
            bool any = false;
            for (int s = 0; s < ns; s++) {
                if (pos[s] + 1 >= cuts[s].size() || rng() % 7 == 0) continue;
                size_t take = 1 + rng() % cfg.max_frames_per_batch;
                if (rng() % 9 == 0) take += 3;  // sometimes more than the decoder was sized for
                size_t e = std::min(pos[s] + take, cuts[s].size() - 1);
                bufs[s] = data[s].data() + cuts[s][pos[s]];
                lens[s] = cuts[s][e] - cuts[s][pos[s]];
                if (damage && rng() % 11 == 0 && lens[s] > 8) lens[s] -= 1 + rng() % 7;  // a chunk cut short
                pos[s] = e;
                any = true;
            }
            if (!any) break;
            h264mi_batch_info info;
            int32_t r = h264mi_batch_prepare(dec, ns, bufs.data(), lens.data(), &info);
            batches++;
            if (r == 0) {
                h264mi_batch_execute(dec);
                if (rng() % 3) h264mi_batch_sync(dec);
            }
            for (int s = 0; s < ns; s++) {
                int32_t st = 0, nf = 0;
                h264mi_stream_status(dec, s, &st);
                if (st == 0) ok_streams++; else failed_streams++;
                codes[st <= 0 && st > -64 ? -st : 63]++;
                if (h264mi_stream_frame_count(dec, s, &nf) != 0) continue;
                frames += nf;
                int32_t order[64], no = 0;
                h264mi_stream_output_order(dec, s, order, 64, &no);
                for (int f = 0; f < nf + 1; f++) {  // one past the end on purpose
                    h264mi_frame_info fi;
                    if (h264mi_frame_get_info(dec, s, f, &fi) != 0) continue;
                    if (rng() % 4 == 0) h264mi_frame_read(dec, s, f, rng() & 1, pix.data(), rng() % 5 ? pix.size() : 1000);
                }
                if (rng() % 13 == 0) h264mi_stream_reset(dec, s);
            }
            if (rng() % 29 == 0) h264mi_decoder_reset(dec);
        }
        h264mi_decoder_destroy(dec);
    }
}

int main(int argc, char **argv) {
    // H264MI_HOST_THREADS=N: N clients side by side, each with its own decoders (what the library promises: decoders are independent;
    // the interesting run is the ThreadSanitizer build, SAN=thread in tools/host_asan.sh)
    if (argc < 4) { fprintf(stderr, "usage: host_asan iterations seed stream.h264...\n"); return 2; }
    const int iters = atoi(argv[1]);
    const unsigned seed = static_cast<unsigned>(atoi(argv[2]));
    for (int i = 3; i < argc; i++) { streams.push_back(slurp(argv[i])); if (streams.back().empty()) { fprintf(stderr, "cannot read %s\n", argv[i]); return 2; } }
    const char *nt = getenv("H264MI_HOST_THREADS");
    const int threads = nt ? std::max(1, atoi(nt)) : 1;
    if (threads == 1)
        client(iters, seed);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; t++) pool.emplace_back(client, (iters + threads - 1) / threads, seed * 7919u + static_cast<unsigned>(t));
        for (auto &t : pool) t.join();
    }
    printf("host asan: %d decoders (%ld refused), %ld batches, stream results %ld ok / %ld failed, %ld frames\n", iters, refused.load(), batches.load(), ok_streams.load(),
           failed_streams.load(), frames.load());
    printf("stream status histogram:");
    for (int i = 0; i < 64; i++) if (codes[i].load()) printf(" %d:%ld", -i, codes[i].load());
    printf("\n");
    return 0;
}
