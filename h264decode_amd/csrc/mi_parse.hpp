// h264decode_amd/csrc/mi_parse.hpp -- host-side bitstream front end of the product:
// Annex-B scan, NAL header + emulation-prevention removal, SPS / PPS / slice-header parsing.
// (Reference layers L1-L4 and the header half of L5: h264/server.go, h264/nalUnit.go,
// h264/bit_reader.go, h264/sps.go, h264/pps.go, h264/slice.go:835-1048.)
#pragma once
#include <cstddef>
#include <cstdint>
#include "../../include/h264mi.h"

namespace mi {

// MSB-first bit cursor over an RBSP with a CLZ-based Exp-Golomb reader
// (reference: BitReader h264/bit_reader.go:11-17, ue/se/te :62-161).
class BitReader {
  public:
    BitReader(const uint8_t *p, size_t n) : p_(p), nbits_(static_cast<int64_t>(n) * 8) {}
    uint32_t u(int n);
    uint32_t ue();
    int32_t se();
    bool more_rbsp_data() const;
    int64_t pos() const { return pos_; }
    bool overrun() const { return pos_ > nbits_; }

  private:
    uint64_t window(int64_t bitpos) const; // 64 bits starting at bitpos (zero padded)
    const uint8_t *p_;
    int64_t nbits_;
    int64_t pos_ = 0;
};

int annexb_scan(const uint8_t *buf, size_t len, h264mi_nal *out, int cap, int *n);
int nal_parse(const uint8_t *nal, size_t len, h264mi_nal *hdr, uint8_t *rbsp, size_t *rbsp_len);
// RBSP extraction only (header byte(s) skipped by the caller via `skip`)
size_t unescape(const uint8_t *src, size_t n, uint8_t *dst);
int parse_sps(const uint8_t *rbsp, size_t len, h264mi_sps *s);
int parse_pps(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *p);
int parse_pps_ids(const h264mi_sps *sps, const uint8_t *rbsp, size_t len, h264mi_pps *p, uint8_t *ids, size_t cap, size_t *n_ids);
// 8.2.2 (h264/slice.go:134-158, :457-552)
int map_unit_to_slice_group_map(const h264mi_sps *s, const h264mi_pps *p, const uint8_t *ids, size_t n_ids, int cycle, uint8_t *map, size_t cap, size_t *n_out);
int mb_to_slice_group_map(const h264mi_sps *s, const h264mi_pps *p, const uint8_t *ids, size_t n_ids, int cycle, int field_pic, uint8_t *map, size_t cap, size_t *n_out);
int next_mb_address(const uint8_t *map, size_t n_mbs, size_t n);
int parse_slice_header(const h264mi_sps *s, const h264mi_pps *p, int nal_ref_idc, int nal_unit_type, const uint8_t *rbsp, size_t len,
                       h264mi_slice_header *sh);
void set_error(const char *fmt, ...);
const char *last_error();

} // namespace mi
