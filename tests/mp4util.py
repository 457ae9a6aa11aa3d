"""Minimal MP4 (ISO BMFF) -> Annex-B rewrap for the optional third-party real-stream tests.

Not part of the product; used only to feed local sample MP4s (if present) to the oracle /
decoder.  Finds the first 'avc1' track, reads avcC (SPS/PPS, NAL length size) and the sample
tables (stsz/stco/co64/stsc) and emits start-code delimited NAL units."""
import struct


def _boxes(buf, start, end):
    pos = start
    while pos + 8 <= end:
        size, typ = struct.unpack(">I4s", buf[pos:pos + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[pos + 8:pos + 16])[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr:
            break
        yield typ, pos + hdr, pos + size
        pos += size


def _find(buf, start, end, path):
    if not path:
        return start, end
    for typ, s, e in _boxes(buf, start, end):
        if typ == path[0]:
            r = _find(buf, s, e, path[1:])
            if r:
                return r
    return None


def mp4_to_annexb(data: bytes) -> bytes:
    moov = _find(data, 0, len(data), [b"moov"])
    if not moov:
        raise ValueError("no moov")
    for typ, s, e in _boxes(data, *moov):
        if typ != b"trak":
            continue
        stbl = _find(data, s, e, [b"mdia", b"minf", b"stbl"])
        if not stbl:
            continue
        stsd = _find(data, *stbl, [b"stsd"])
        if not stsd:
            continue
        # stsd: version/flags(4) entry_count(4) then sample entries
        ent = stsd[0] + 8
        esize, etype = struct.unpack(">I4s", data[ent:ent + 8])
        if etype != b"avc1":
            continue
        # VisualSampleEntry is 78 bytes after the 8-byte box header
        avcc = _find(data, ent + 8 + 78, ent + esize, [b"avcC"])
        if not avcc:
            continue
        a = data[avcc[0]:avcc[1]]
        nal_len_size = (a[4] & 3) + 1
        out = bytearray()
        pos = 5
        nsps = a[pos] & 31
        pos += 1
        for _ in range(nsps):
            n = struct.unpack(">H", a[pos:pos + 2])[0]
            out += b"\x00\x00\x00\x01" + a[pos + 2:pos + 2 + n]
            pos += 2 + n
        npps = a[pos]
        pos += 1
        for _ in range(npps):
            n = struct.unpack(">H", a[pos:pos + 2])[0]
            out += b"\x00\x00\x00\x01" + a[pos + 2:pos + 2 + n]
            pos += 2 + n
        # sample tables
        stsz = _find(data, *stbl, [b"stsz"])
        ss, cnt = struct.unpack(">II", data[stsz[0] + 4:stsz[0] + 12])
        sizes = [ss] * cnt if ss else list(struct.unpack(">%dI" % cnt, data[stsz[0] + 12:stsz[0] + 12 + 4 * cnt]))
        stco = _find(data, *stbl, [b"stco"])
        if stco:
            n = struct.unpack(">I", data[stco[0] + 4:stco[0] + 8])[0]
            chunks = list(struct.unpack(">%dI" % n, data[stco[0] + 8:stco[0] + 8 + 4 * n]))
        else:
            co64 = _find(data, *stbl, [b"co64"])
            n = struct.unpack(">I", data[co64[0] + 4:co64[0] + 8])[0]
            chunks = list(struct.unpack(">%dQ" % n, data[co64[0] + 8:co64[0] + 8 + 8 * n]))
        stsc = _find(data, *stbl, [b"stsc"])
        n = struct.unpack(">I", data[stsc[0] + 4:stsc[0] + 8])[0]
        runs = [struct.unpack(">III", data[stsc[0] + 8 + 12 * i:stsc[0] + 20 + 12 * i]) for i in range(n)]
        si = 0
        for ci, coff in enumerate(chunks):
            spc = 0
            for first, per, _ in runs:
                if ci + 1 >= first:
                    spc = per
            off = coff
            for _ in range(spc):
                if si >= len(sizes):
                    break
                end = off + sizes[si]
                p = off
                while p + nal_len_size <= end:
                    n = int.from_bytes(data[p:p + nal_len_size], "big")
                    p += nal_len_size
                    out += b"\x00\x00\x00\x01" + data[p:p + n]
                    p += n
                off = end
                si += 1
        return bytes(out)
    raise ValueError("no avc1 track")
